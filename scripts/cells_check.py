#!/usr/bin/env python3
"""Diagnostic (GPU): the stored-cell census (REC_CELLS) of every forward kernel on the same batch."""
import sys, numpy as np
sys.path.insert(0, ".")
import wfa_amd as w
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
data = w.generate_pairs(3, n, 1000, 0.05, n_threads=16)
for name, opts in (("blk16", {"blk": 16}), ("blk8", {"blk": 8}), ("reg", {"blk": 0}), ("packed", {"blk": 0, "reg": 0}), ("generic", {"packed": 0})):
    al = w.New(); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    for k, v in opts.items(): al.set_option(k, v)
    r = al.align_arrays(*data)
    t = al.last_timing()
    print(name, "cells_stored", t.cells_stored, "retried", t.n_retried_pairs, "kind", t.main_kernel_kind)
    al.close()
