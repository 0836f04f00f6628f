#!/usr/bin/env python3
"""Diagnostic (GPU): phases of the host entries (bytes / pre-packed) with WFAHIP_DEBUG_TIMING=1, C-ABI called directly."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, ".")
os.environ["WFAHIP_DEBUG_TIMING"] = "1"
import wfa_amd as w
from wfa_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
blob, q_off, q_len, t_off, t_len = w.generate_pairs(3, n, 1000, 0.05, n_threads=32)
packed, q_woff, t_woff = w.pack_pairs(blob, q_off, q_len, t_off, t_len, n_threads=32)
al = w.New(); al.AdaptiveReduction(w.DefaultAdaptiveOption)
prm = al._params(); vp = lambda a: a.ctypes.data_as(C.c_void_p)
for name in ("bytes", "packed", "bytes", "packed"):
    for it in range(3):
        res = L.Results(); t0 = time.perf_counter()
        if name == "bytes":
            rc = L.lib().wfahip_align_batch(al._ctx, C.byref(prm), vp(blob), blob.size, vp(q_off), vp(q_len), vp(t_off), vp(t_len), n, C.byref(res))
        else:
            rc = L.lib().wfahip_align_batch_packed(al._ctx, C.byref(prm), vp(packed), packed.size, vp(q_woff), vp(q_len), vp(t_woff), vp(t_len), n, C.byref(res))
        dt = time.perf_counter() - t0
        L.lib().wfahip_results_free(C.byref(res))
        print(f"== {name} call {it}: rc {rc} {dt*1e3:.1f} ms", file=sys.stderr, flush=True)
