"""GPU box: a 3e6-pair batch (several chunks) with and without the streamed backtrace; prints the library's time per call."""
import os, sys, time
os.environ["WFAHIP_NO_UPLOAD_OVERLAP"] = "1"
sys.path.insert(0, ".")
import wfa_amd as w
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
data = w.generate_pairs(seed=201, n_pairs=n, length=1000, error_rate=0.05, n_threads=32)
for opts in ({}, {"bt_stream": 0}, {"bt_stream": 0, "overlap": 1}):
    al = w.New(); al.AdaptiveReduction(w.DefaultAdaptiveOption)
    for k, v in opts.items(): al.set_option(k, v)
    ts = []
    for rep in range(3):
        al.align_arrays(*data); t = al.last_timing(); ts.append(t.total_ms)
    print(opts, "device ms per call", [round(x, 1) for x in ts], "forward launches", t.n_main_launches, "main kernel ms", round(t.main_kernel_ms, 1), flush=True)
    al.close()
