#!/usr/bin/env python3
"""GPU box: streamed backtrace (option bt_stream) against the default pipeline on one batch -- every result array
and every CIGAR op must be identical.  Usage: scripts/stream_check.py [pairs] [length] [waves]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import wfa_amd as w

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 96
data = w.generate_pairs(11, n, length, 0.05, n_threads=32)
res = {}
for name, opt in (("base", 0), ("stream", waves)):
    al = w.New()
    al.AdaptiveReduction(w.DefaultAdaptiveOption)
    al.set_option("bt_stream", opt)
    al.set_option("bt_stream_single", 1)
    al.set_option("bt_stream_min", 1)
    al.align_arrays(*data)
    t0 = time.perf_counter()
    r = al.align_arrays(*data)
    dt = time.perf_counter() - t0
    t = al.last_timing()
    print(name, "host ms", round(dt * 1e3, 2), "lib total ms", round(t.total_ms, 3), "fwd ms", round(t.main_kernel_ms, 3),
          "retried", t.n_retried_pairs, "cells", t.cells_stored)
    res[name] = r
    al.close()
a, b = res["base"], res["stream"]
bad = 0
for f in ("status", "score", "tbegin", "tend", "qbegin", "qend", "align_len", "matches", "gaps", "gap_regions", "ops_len"):
    if not np.array_equal(getattr(a, f), getattr(b, f)):
        print("DIFF in", f, int((getattr(a, f) != getattr(b, f)).sum()))
        bad += 1
if not np.array_equal(a.ops, b.ops):
    print("DIFF in ops")
    bad += 1
print("identical" if bad == 0 else "MISMATCH")
sys.exit(1 if bad else 0)
