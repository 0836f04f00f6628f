"""GPU: BASELINE configs[3] (pair-sharded 1e7 x 1 kbp @5 %) as far as a 1-GPU box allows.

(a) bench.py's own multi-rank branch with REAL kernels: `--gpus 2 --share-gpus` starts two ranks that share the one GPU,
    each aligns its shard of the seed-4 dataset (shard_range + the generator's first_index), the records are gathered
    onto rank 0 (gloo through host copies: two RCCL ranks cannot share a device) -- and the gathered records of BOTH
    shards must equal the oracle's, in pair order.
(b) the full 1e7-pair workload on one GPU (generated in HBM, several chunks): size-independent properties on every pair
    plus the oracle on 1e5 pairs drawn from 100 places across the dataset (wfa.go:73-78 is the model being sharded:
    one aligner per worker, pairs independent).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.timeout(900)
@pytest.mark.parametrize("total", [20001])
def test_two_ranks_share_one_gpu_against_oracle(built, tmp_path, total):
    import wfa_amd as w
    from wfa_amd import _lib as L
    from oracle import oracle as O
    dump = str(tmp_path / "records.npy")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpus", "--config", "c4",
                        "--total-pairs", str(total), "--steps", "2", "--warmup", "1", "--dump-records", dump],
                       env=_clean_env(), capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    c = out["config"]
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and c["backend"] == "gloo"
    assert c["pairs_per_rank"] == [total - total // 2, total // 2] and c["gathered_records_complete"] is True
    assert c["gpus_shared"] and "roofline" in out and out["roofline"]["kernel_ms"] > 0  # the kernels ran
    rec = np.load(dump).view(np.uint32)
    assert rec.shape == (total, L.REC_OPS_OFF_LO)
    data = w.generate_pairs(seed=4, n_pairs=total, length=1000, error_rate=0.05)
    want = O.align_batch(O.make_params(adaptive=(10, 50, 1)), *data, n_threads=max(8, (os.cpu_count() or 8) // 2),
                         want_ops=True)
    assert (rec[:, L.REC_STATUS] == 0).all()
    for f, col in (("score", L.REC_SCORE), ("tbegin", L.REC_TBEGIN), ("tend", L.REC_TEND), ("qbegin", L.REC_QBEGIN),
                   ("qend", L.REC_QEND), ("align_len", L.REC_ALIGN_LEN), ("matches", L.REC_MATCHES), ("gaps", L.REC_GAPS),
                   ("gap_regions", L.REC_GAP_REGIONS), ("ops_len", L.REC_OPS_LEN)):
        a, b = rec[:, col].astype(np.int64), getattr(want, f).astype(np.int64)
        assert np.array_equal(a, b), (f, np.nonzero(a != b)[0][:5])


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_one_rank_rccl_gather_against_oracle(built, tmp_path):
    """The RCCL branch of the result gather, executed: ONE rank with backend `nccl` on the GPU (tests/nccl_one_rank.py, a
    child process) gathers the records -- and, second form, the CIGAR op arrays -- of a real 2 000-pair alignment from
    device tensors; what rank 0 received must be the oracle's results (the pairs' ops addressed through the gathered
    OPS_OFF fields).  The sharding model: wfa.go:73-78 (one aligner per worker, pairs independent)."""
    import wfa_amd as w
    from wfa_amd import _lib as L
    from oracle import oracle as O
    n, seed = 2000, 44
    out = str(tmp_path / "gathered.npz")
    env = _clean_env()
    env["MASTER_PORT"] = str(_free_port())
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_one_rank.py"), out, str(n), str(seed)],
                       env=env, capture_output=True, text=True, timeout=550)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    g = np.load(out)
    data = w.generate_pairs(seed=seed, n_pairs=n, length=1000, error_rate=0.05)
    want = O.align_batch(O.make_params(adaptive=(10, 50, 1)), *data, n_threads=8, want_ops=True)
    assert g["rec_only"].shape == (n, L.REC_OPS_OFF_LO) and g["rec_full"].shape == (n, L.REC_WORDS)
    assert int(g["ops_only_len"]) == 0 and int(g["census"][0]) == n and float(g["reduced"][0]) == 1.25
    for rec in (g["rec_only"].view(np.uint32), g["rec_full"].view(np.uint32)):
        assert (rec[:, L.REC_STATUS] == 0).all()
        for f, col in (("score", L.REC_SCORE), ("tbegin", L.REC_TBEGIN), ("tend", L.REC_TEND), ("qbegin", L.REC_QBEGIN),
                       ("qend", L.REC_QEND), ("align_len", L.REC_ALIGN_LEN), ("matches", L.REC_MATCHES), ("gaps", L.REC_GAPS),
                       ("gap_regions", L.REC_GAP_REGIONS), ("ops_len", L.REC_OPS_LEN)):
            assert np.array_equal(rec[:, col].astype(np.int64), getattr(want, f).astype(np.int64)), f
    rec, ops = g["rec_full"].view(np.uint32), g["ops"].view(np.uint64)
    assert ops.shape[0] == int(g["n_ops"])
    off = rec[:, L.REC_OPS_OFF_LO].astype(np.uint64) | (rec[:, L.REC_OPS_OFF_HI].astype(np.uint64) << np.uint64(32))
    for i in range(n):
        assert np.array_equal(ops[int(off[i]):int(off[i]) + int(rec[i, L.REC_OPS_LEN])], want.pair_ops(i)), i


@pytest.mark.timeout(900)
def test_bench_force_collective_runs_the_nccl_lines(built, tmp_path):
    """bench.py --gpus 1 --force-collective: the process group is RCCL with one rank, and every collective of the multi-GPU
    path (barriers, the asynchronous record gather beside the next step, the max-over-ranks reduction, the rank census) runs
    on device tensors; the gathered records equal the oracle's."""
    import wfa_amd as w
    from wfa_amd import _lib as L
    from oracle import oracle as O
    total = 20001
    dump = str(tmp_path / "records.npy")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--config", "c4",
                        "--total-pairs", str(total), "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--host-entry", "0",
                        "--latency", "0", "--dump-records", dump], env=_clean_env(), capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    c = out["config"]
    assert out["n_gpus"] == 1 and c["backend"] == "nccl" and c["collective_forced"] is True
    assert c["pairs_per_rank"] == [total] and c["gathered_records_complete"] is True and c["gather_ms_standalone"] > 0
    assert c["gather_bytes_per_rank_step"] == total * L.REC_OPS_OFF_LO * 4
    rec = np.load(dump).view(np.uint32)
    assert rec.shape == (total, L.REC_OPS_OFF_LO)
    data = w.generate_pairs(seed=4, n_pairs=total, length=1000, error_rate=0.05)
    want = O.align_batch(O.make_params(adaptive=(10, 50, 1)), *data, n_threads=max(8, (os.cpu_count() or 8) // 2), want_ops=False)
    assert (rec[:, L.REC_STATUS] == 0).all()
    for f, col in (("score", L.REC_SCORE), ("tbegin", L.REC_TBEGIN), ("tend", L.REC_TEND), ("qbegin", L.REC_QBEGIN),
                   ("qend", L.REC_QEND), ("align_len", L.REC_ALIGN_LEN), ("matches", L.REC_MATCHES), ("gaps", L.REC_GAPS),
                   ("gap_regions", L.REC_GAP_REGIONS)):
        assert np.array_equal(rec[:, col].astype(np.int64), getattr(want, f).astype(np.int64)), f


@pytest.mark.timeout(1500)
def test_config4_full_count_on_one_gpu(built):
    """configs[3]'s 1e7 x 1 kbp pairs, one GPU: every pair OK; CIGAR cost == score and both sequences consumed (computed on
    the device over all ~9e8 ops); a second pass is bit-identical; 100 runs of 1 000 consecutive pairs spread over the
    dataset -- every chunk of the pass is hit -- equal the oracle on every field and every op."""
    import ctypes as C
    import torch
    import wfa_amd as w
    from wfa_amd import _lib as L
    from oracle import oracle as O
    n, length, err, seed = 10_000_000, 1000, 0.05, 4
    dev = torch.device("cuda:0")
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=True), device=0)
    assert al.AdaptiveReduction(w.DefaultAdaptiveOption) is None
    blob, q_off, q_len, t_off, t_len = w.generate_pairs_device(al, seed, n, length, err)
    max_len = int(max(q_len.max().item(), t_len.max().item()))
    sum_len = int(q_len.sum().item() + t_len.sum().item())
    ops_cap = sum_len // 4 + 8 * n + 1024
    d_rec = torch.zeros((n, L.REC_WORDS), dtype=torch.int32, device=dev)
    d_ops = torch.zeros(ops_cap, dtype=torch.int64, device=dev)
    prm = al._params()
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run():
        needed = C.c_uint64()
        L.check(L.lib().wfahip_align_batch_device(al._ctx, C.byref(prm), blob.data_ptr(), blob.numel(), q_off.data_ptr(),
                                                  q_len.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n, max_len,
                                                  d_rec.data_ptr(), d_ops.data_ptr(), ops_cap, C.byref(needed), stream),
                "wfahip_align_batch_device")
        torch.cuda.synchronize(dev)
        return int(needed.value)

    n_ops = run()
    rec1 = d_rec[:, :L.REC_CELLS_LO].clone()
    ops_off = (d_rec[:, L.REC_OPS_OFF_LO].to(torch.int64) & 0xFFFFFFFF) | (d_rec[:, L.REC_OPS_OFF_HI].to(torch.int64) << 32)
    ops_len = d_rec[:, L.REC_OPS_LEN].to(torch.int64)
    assert bool((d_rec[:, L.REC_STATUS] == 0).all())
    assert int(ops_len.sum().item()) <= n_ops <= ops_cap  # (n_ops: the op buffer's high-water mark, regions included)
    # ---- properties over every op of every pair, on the device, a million pairs at a time (a pair's ops are one
    # contiguous run of the op buffer, at ops_off)
    order = torch.argsort(ops_off)
    so, sl = ops_off[order], ops_len[order]
    assert bool((so[1:] >= so[:-1] + sl[:-1]).all())  # no two pairs' op lists overlap
    assert int((so[-1] + sl[-1]).item()) <= ops_cap
    del order, so, sl
    n_overshoot = n_merged = 0
    for a in range(0, n, 1_000_000):
        b = min(n, a + 1_000_000)
        ln = ops_len[a:b]
        tot = int(ln.sum().item())
        pidu = torch.repeat_interleave(torch.arange(b - a, device=dev), ln)
        run_start = torch.cumsum(ln, 0) - ln
        pos = ops_off[a:b][pidu] + (torch.arange(tot, device=dev) - run_start[pidu])
        ops = d_ops[pos]
        lu, cu = (ops >> 32) & 0xFF, ops & 0xFFFFFFFF
        isM, isX, isI, isD, isH = (lu == ord("M")), (lu == ord("X")), (lu == ord("I")), (lu == ord("D")), (lu == ord("H"))
        assert bool((isM | isX | isI | isD | isH).all()) and bool((cu > 0).all())
        z = lambda: torch.zeros(b - a, dtype=torch.int64, device=dev)
        q_used = z().index_add_(0, pidu, cu * (isM | isX | isD | isH))
        t_used = z().index_add_(0, pidu, cu * (isM | isX | isI))
        cost = z().index_add_(0, pidu, isX * cu * 4 + (isI | isD | isH) * (6 + 2 * cu))
        dq, dt = q_used - q_len[a:b].to(torch.int64), t_used - t_len[a:b].to(torch.int64)
        assert bool((dq.abs() <= 1).all()) and bool((dt.abs() <= 1).all())
        n_overshoot += int(((dq != 0) | (dt != 0)).sum().item())
        # cost of the CIGAR == score, except where process() merges two gaps that were opened separately into one op
        # (wfa_cigar.go:136-214: "1I" + "1I" -> "2I" reads as one gap of cost 10, the alignment paid 16): 2-3 pairs per
        # million, bit for bit the oracle's CIGARs (checked on the pairs in question)
        sc = d_rec[a:b, L.REC_SCORE].to(torch.int64) & 0xFFFFFFFF
        assert bool((cost <= sc).all())
        n_merged += int((cost != sc).sum().item())
        # merged ops: no two neighbours of a pair's list carry the same letter (wfa_cigar.go:136-214)
        same = (lu[1:] == lu[:-1]) & (pidu[1:] == pidu[:-1])
        assert not bool(same.any())
        del pidu, run_start, pos, ops, lu, cu, q_used, t_used, cost, dq, dt, same
    assert n_overshoot <= n // 10_000  # the reference's own off-by-one overshoot, about 1 pair in 2e5
    assert n_merged <= n // 100_000
    # ---- the oracle on 100 runs of 1 000 pairs across the dataset
    rec_h = d_rec.cpu().numpy().view(np.uint32)
    thr = max(8, (os.cpu_count() or 8) // 2)
    p = O.make_params(adaptive=(10, 50, 1))
    for r in range(100):
        first = r * (n // 100) + (r * 7919) % (n // 100 - 1000)
        data = w.generate_pairs(seed=seed, n_pairs=1000, length=length, error_rate=err, first_index=first)
        want = O.align_batch(p, *data, n_threads=thr)
        sl_ = slice(first, first + 1000)
        for f, col in (("score", L.REC_SCORE), ("tbegin", L.REC_TBEGIN), ("tend", L.REC_TEND), ("qbegin", L.REC_QBEGIN),
                       ("qend", L.REC_QEND), ("align_len", L.REC_ALIGN_LEN), ("matches", L.REC_MATCHES),
                       ("gaps", L.REC_GAPS), ("gap_regions", L.REC_GAP_REGIONS), ("ops_len", L.REC_OPS_LEN)):
            assert np.array_equal(rec_h[sl_, col].astype(np.int64), getattr(want, f).astype(np.int64)), (r, f)
        ln = ops_len[first:first + 1000]
        pidu = torch.repeat_interleave(torch.arange(1000, device=dev), ln)
        pos = ops_off[first:first + 1000][pidu] + (torch.arange(int(ln.sum().item()), device=dev) - (torch.cumsum(ln, 0) - ln)[pidu])
        assert np.array_equal(d_ops[pos].cpu().numpy().view(np.uint64), want.ops), (r, "CIGAR ops")
    # ---- determinism: a second pass gives the same records (op offsets may differ between passes: compare the fields)
    run()
    assert bool(torch.equal(d_rec[:, :L.REC_OPS_OFF_LO], rec1[:, :L.REC_OPS_OFF_LO]))
    w.RecycleAligner(al)


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("config,sample", [("k10", 20_000), ("k20", 8_000), ("l5", 64), ("l10", 48), ("l20", 24)])
def test_reference_grid_configs_against_oracle(built, config, sample):
    """The reference's published grid (README.md:326-345: 1e5 x 1 kbp and 500 x 50 kbp at 5 / 10 / 20 % error, global,
    wf-adaptive 10/50/1) as bench.py runs it (--config k10 | k20 | l5 | l10 | l20): the first `sample` pairs of each
    dataset, every field and every CIGAR op against the oracle."""
    sys.path.insert(0, ROOT)
    import bench
    import wfa_amd as w
    from oracle import oracle as O
    from test_parity_gpu import _aligner, _oracle_params, assert_batch_equal
    c = bench.CONFIGS[config]
    data = w.generate_pairs(seed=c["seed"], n_pairs=sample, length=c["length"], error_rate=c["error"])
    al = _aligner(True, (10, 50, 1))
    got = al.align_arrays(*data)
    want = O.align_batch(_oracle_params(True, (10, 50, 1)), *data, n_threads=max(8, (os.cpu_count() or 8) // 2))
    assert_batch_equal(got, want, config)
    again = al.align_arrays(*data)
    assert_batch_equal(again, want, config + " (second call)")
    al.close()
