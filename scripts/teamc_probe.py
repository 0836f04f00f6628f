"""Diagnostic (GPU box): small pairs through wfa_teamc_kernel, one configuration at a time, with the library's own account of
a failure (wfahip_last_error) and the first difference against the oracle.  Usage: python scripts/teamc_probe.py [case ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("WFAHIP_DEBUG_TIMING", "1")
import numpy as np
import wfa_amd as w
from wfa_amd import _lib as L
from oracle import oracle as O

CASES = {
    # name: (n_pairs, length, err, glob, ad, opts)
    "g_T2_team": (4, 300, 0.08, True, (10, 50, 1), {"team_wgs": 2, "team_solo_max": 0}),
    "g_T1": (4, 300, 0.08, True, (10, 50, 1), {"team_wgs": 1, "team_solo_max": 4096, "team_wave": 0}),
    "g_T2_solo": (4, 300, 0.08, True, (10, 50, 1), {"team_wgs": 2, "team_solo_max": 4096, "team_wave": 0}),
    "g_T2_wave": (4, 300, 0.08, True, (10, 50, 1), {"team_wgs": 2, "team_solo_max": 4096, "team_wave": 1}),
    "s_T2_team": (4, 300, 0.08, False, (10, 50, 1), {"team_wgs": 2, "team_solo_max": 0}),
    "s_T2_wave": (4, 300, 0.08, False, (10, 50, 1), {"team_wgs": 2, "team_solo_max": 64, "team_wave": 1}),
    "s_T3_xbuf": (2, 7000, 0.08, False, (10, 50, 1), {"team_wgs": 3, "team_solo_max": 64, "team_wave": 1}),
    "s_big": (2, 40000, 0.10, False, (10, 50, 1), {}),
}
names = sys.argv[1:] or list(CASES)
for name in names:
    n, length, err, glob, ad, opts = CASES[name]
    data = w.generate_pairs(seed=7, n_pairs=n, length=length, error_rate=err)
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=glob))
    if ad:
        al.AdaptiveReduction(w.AdaptiveReductionOption(*ad))
    for k, v in dict({"packed": 0, "team_min_len": 1, "arena_poison": 1}, **opts).items():
        al.set_option(k, v)
    t0 = time.perf_counter()
    try:
        got = al.align_arrays(*data)
    except Exception as e:  # noqa: BLE001
        print(f"[{name}] FAILED after {time.perf_counter() - t0:.2f} s: {e}; last_error = {L.lib().wfahip_last_error(al._ctx)!r}", flush=True)
        continue
    dt = time.perf_counter() - t0
    want = O.align_batch(O.make_params(global_alignment=glob, adaptive=ad), *data, n_threads=4)
    bad = [f for f in ("status", "score", "tbegin", "tend", "qbegin", "qend", "align_len", "matches", "gaps", "gap_regions", "ops_len")
           if not np.array_equal(getattr(got, f), getattr(want, f))]
    ops_ok = all(np.array_equal(got.pair_ops(i), want.pair_ops(i)) for i in range(n))
    t = al.last_timing()
    print(f"[{name}] {dt:.2f} s kind {t.main_kernel_kind} launches {t.n_launches} retried {t.n_retried_pairs}: "
          f"{'OK' if not bad and ops_ok else 'DIFFERS in ' + str(bad) + ('' if ops_ok else ' + ops')}; status {got.status[:4]} score {got.score[:4]} want {want.score[:4]}", flush=True)
    al.close()
