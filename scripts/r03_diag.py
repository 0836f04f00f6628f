import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import wfa_amd as w
from wfa_amd import _lib as L
from oracle import oracle as O
n, length, err, seed = int(sys.argv[1]), 1000, 0.05, 4
dev = torch.device("cuda:0")
al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=True), device=0)
al.AdaptiveReduction(w.DefaultAdaptiveOption)
blob, q_off, q_len, t_off, t_len = w.generate_pairs_device(al, seed, n, length, err)
max_len = int(max(q_len.max().item(), t_len.max().item()))
sum_len = int(q_len.sum().item() + t_len.sum().item())
ops_cap = sum_len // 4 + 8 * n + 1024
prm = al._params()
stream = torch.cuda.current_stream(dev).cuda_stream
def run(duo):
    al.set_option("duo", duo)
    d_rec = torch.zeros((n, L.REC_WORDS), dtype=torch.int32, device=dev)
    d_ops = torch.zeros(ops_cap, dtype=torch.int64, device=dev)
    needed = C.c_uint64()
    L.check(L.lib().wfahip_align_batch_device(al._ctx, C.byref(prm), blob.data_ptr(), blob.numel(), q_off.data_ptr(), q_len.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n, max_len, d_rec.data_ptr(), d_ops.data_ptr(), ops_cap, C.byref(needed), stream), "align")
    torch.cuda.synchronize(dev)
    t = al.last_timing()
    print("duo", duo, "kind", t.main_kernel_kind, "launches", t.n_launches, "main launches", t.n_main_launches, "retried", t.n_retried_pairs, "ms", t.total_ms)
    return d_rec, d_ops
r1, o1 = run(1)
r0, o0 = run(0)
same = (r1[:, :L.REC_OPS_OFF_LO] == r0[:, :L.REC_OPS_OFF_LO]).all(dim=1)
bad = torch.nonzero(~same).flatten()
print("records differing between duo=1 and duo=0:", bad.numel(), bad[:20].tolist())
for i in bad[:5].tolist():
    print(i, r1[i, :11].tolist(), r0[i, :11].tolist())
    data = w.generate_pairs(seed=seed, n_pairs=1, length=length, error_rate=err, first_index=i)
    want = O.align_batch(O.make_params(adaptive=(10, 50, 1)), *data, n_threads=1)
    print("   oracle score", want.score[0], "ops_len", want.ops_len[0], "region", want.tbegin[0], want.tend[0], want.qbegin[0], want.qend[0])
# cost check on duo=0 output
def costs(d_rec, d_ops):
    ops_off = (d_rec[:, L.REC_OPS_OFF_LO].to(torch.int64) & 0xFFFFFFFF) | (d_rec[:, L.REC_OPS_OFF_HI].to(torch.int64) << 32)
    ops_len = d_rec[:, L.REC_OPS_LEN].to(torch.int64)
    badp = []
    for a in range(0, n, 1_000_000):
        b = min(n, a + 1_000_000)
        ln = ops_len[a:b]; tot = int(ln.sum().item())
        pidu = torch.repeat_interleave(torch.arange(b - a, device=dev), ln)
        pos = ops_off[a:b][pidu] + (torch.arange(tot, device=dev) - (torch.cumsum(ln, 0) - ln)[pidu])
        ops = d_ops[pos]; lu, cu = (ops >> 32) & 0xFF, ops & 0xFFFFFFFF
        isX, isG = (lu == ord("X")), ((lu == ord("I")) | (lu == ord("D")) | (lu == ord("H")))
        cost = torch.zeros(b - a, dtype=torch.int64, device=dev).index_add_(0, pidu, isX * cu * 4 + isG * (6 + 2 * cu))
        bp = torch.nonzero(cost != (d_rec[a:b, L.REC_SCORE].to(torch.int64) & 0xFFFFFFFF)).flatten() + a
        badp += bp.tolist()
    return badp
for name, (r, o) in (("duo1", (r1, o1)), ("duo0", (r0, o0))):
    bp = costs(r, o)
    print(name, "pairs with CIGAR cost != score:", len(bp), bp[:10])
    for i in bp[:3]:
        data = w.generate_pairs(seed=seed, n_pairs=1, length=length, error_rate=err, first_index=i)
        want = O.align_batch(O.make_params(adaptive=(10, 50, 1)), *data, n_threads=1)
        off = (int(r[i, L.REC_OPS_OFF_LO]) & 0xFFFFFFFF) | (int(r[i, L.REC_OPS_OFF_HI]) << 32)
        got = o[off:off + int(r[i, L.REC_OPS_LEN])].cpu().numpy().view(np.uint64)
        print("  pair", i, "gpu score", int(r[i, L.REC_SCORE]), "oracle", int(want.score[0]), "ops equal:", np.array_equal(got, want.pair_ops(0)), "oracle cigar", want.cigar(0)[:80], "q_len t_len", int(q_len[i]), int(t_len[i]))
