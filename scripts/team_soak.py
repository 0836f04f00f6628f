#!/usr/bin/env python3
"""Soak of the team kernel (GPU): the configs[4] sample (8 x 100 kbp semi-global pairs) aligned `reps` times over a poisoned
pool with the given options; every pass must reproduce the first one bit for bit (the first pass of the default options is
what tests/test_parity_gpu.py::test_config5_full_length_pair checks against the oracle).  Usage: team_soak.py reps key=value ..."""
import os; os.environ.setdefault("WFAHIP_DEBUG", "1")  # (these are debug / experiment knobs)
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
torch.zeros(1, device="cuda:0")
import wfa_amd as w
reps = int(sys.argv[1])
opts = dict(kv.split("=") for kv in sys.argv[2:])
data = w.generate_pairs(seed=5, n_pairs=8, length=100000, error_rate=0.10, n_threads=8)
ref = None
for label, o in (("default", {}), ("options", opts)):
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=False)); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    al.set_option("arena_poison", 1)
    for k, v in o.items():
        al.set_option(k, int(v))
    bad, t0 = 0, time.time()
    for r in range(reps if o else 2):
        got = al.align_arrays(*data)
        if ref is None:
            ref = got
            continue
        same = all(np.array_equal(getattr(got, f), getattr(ref, f)) for f in ("status", "score", "tbegin", "tend", "qbegin", "qend", "align_len", "matches", "gaps", "gap_regions", "ops_len", "ops"))
        bad += not same
    print(f"{label} {o}: {reps if o else 2} passes, {bad} deviations, {time.time() - t0:.1f} s, kernel {al.last_timing().kernel_ms:.0f} ms", flush=True)
    al.close()
