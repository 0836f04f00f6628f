#!/usr/bin/env python3
"""Time the device-resident entry under several option sets (interleaved rounds in ONE process)."""
import sys, os, time, ctypes as C, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wfa_amd as w
from wfa_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
variants = [eval(a) for a in sys.argv[2:]] or [dict()]
data = w.generate_pairs(3, n, 1000, 0.05, n_threads=32)
blob, q_off, q_len, t_off, t_len = data
dev = torch.device("cuda:0")
d = [torch.from_numpy(x.view(np.uint8) if x.dtype == np.uint8 else x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32)).to(dev)
     for x in (blob, q_off, q_len, t_off, t_len)]
ops_cap = int(q_len.sum() + t_len.sum()) // 4 + 8 * n + 1024
rec = torch.empty((n, 16), dtype=torch.int32, device=dev); ops = torch.empty(ops_cap, dtype=torch.int64, device=dev)
al = w.New(); al.AdaptiveReduction(w.DefaultAdaptiveOption); prm = al._params(); lib = L.lib(); tm = L.Timing()
def run():
    need = C.c_uint64()
    L.check(lib.wfahip_align_batch_device(al._ctx, C.byref(prm), d[0].data_ptr(), blob.size, d[1].data_ptr(), d[2].data_ptr(),
            d[3].data_ptr(), d[4].data_ptr(), n, int(max(q_len.max(), t_len.max())), rec.data_ptr(), ops.data_ptr(), ops_cap, C.byref(need), None))
    lib.wfahip_last_timing(al._ctx, C.byref(tm))
defaults = dict(overlap=0, packed_waves_per_cu=0, chunk_pairs=0, packed=1, reg=1)
res = {i: [] for i in range(len(variants))}
ref = None
for rnd in range(4):
    for i, v in enumerate(variants):
        for k, val in {**defaults, **v}.items(): al.set_option(k, val)
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rnd: res[i].append((dt * 1e3, tm.main_kernel_ms, tm.kernel_ms, tm.n_main_launches))
        r = rec.cpu().numpy()[:, :11]
        if ref is None: ref = r.copy()
        assert np.array_equal(ref, r), "results changed with options!"
for i, v in enumerate(variants):
    a = np.array(res[i])
    print(f"{str(v):60s} wall ms median {np.median(a[:,0]):7.2f} min {a[:,0].min():7.2f} | fwd {np.median(a[:,1]):6.2f} all-kernels {np.median(a[:,2]):6.2f} chunks {int(a[0,3])}  -> {n/np.median(a[:,0])*1e3/1e6:.2f} Mpairs/s")
