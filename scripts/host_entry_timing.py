#!/usr/bin/env python3
"""Diagnostic (GPU): phases of wfahip_align_batch on the headline batch (WFAHIP_DEBUG_TIMING=1 prints them to stderr)."""
import os, sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
torch.zeros(1, device="cuda:0")
import wfa_amd as w
from wfa_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
blob, q_off, q_len, t_off, t_len = w.generate_pairs(3, n, 1000, 0.05, n_threads=32)
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=True)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
prm = al._params(); vp = lambda a: a.ctypes.data_as(C.c_void_p)
for i in range(5):
    res = L.Results(); t1 = time.perf_counter()
    L.check(L.lib().wfahip_align_batch(al._ctx, C.byref(prm), vp(blob), blob.size, vp(q_off), vp(q_len), vp(t_off), vp(t_len), n, C.byref(res)))
    dt = time.perf_counter() - t1
    L.lib().wfahip_results_free(C.byref(res))
    print(f"call {i}: {dt * 1e3:.1f} ms", file=sys.stderr, flush=True)
