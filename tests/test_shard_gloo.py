"""CPU, world_size 2 over gloo: the multi-GPU path's sharding + result gather (bench.py uses the same
functions over RCCL).  Records are produced by the ORACLE here (no GPU in this container) -- the point is the
exchange logic: contiguous shards, uneven sizes, op offsets staying valid per rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import wfa_amd as w
    from wfa_amd.shard import gather_results, gather_results_async, shard_range
    from oracle import oracle as O

    b, e = shard_range(n_total, rank, world)
    data = w.generate_pairs(seed=9, n_pairs=e - b, length=120, error_rate=0.05, first_index=b, n_threads=1)
    r = O.align_batch(O.make_params(adaptive=(10, 50, 1)), *data)
    n = e - b
    rec = np.zeros((n, 16), dtype=np.int32)
    rec[:, 1] = r.score
    rec[:, 10] = r.ops_len
    rec[:, 11] = (r.ops_off & 0xFFFFFFFF).astype(np.int64).astype(np.int32)
    ops = np.zeros(len(r.ops) + 17, dtype=np.int64)  # buffer larger than the used prefix, like the device one
    ops[:len(r.ops)] = r.ops.view(np.int64)
    t_rec, t_ops = torch.from_numpy(rec.copy()), torch.from_numpy(ops.copy())
    pend = gather_results_async(t_rec, t_ops, len(r.ops), dst=0)  # bench.py's form: buffers are reused at once
    t_rec.zero_(), t_ops.zero_()
    got = pend.wait()
    again = gather_results(torch.from_numpy(rec), torch.from_numpy(ops), len(r.ops), dst=0)
    only_rec = gather_results(torch.from_numpy(rec), torch.from_numpy(ops), len(r.ops), dst=0, with_ops=False)
    if rank == 0:
        assert all(torch.equal(a, b) for a, b in zip(got[0], again[0]))
        assert all(torch.equal(a, b) for a, b in zip(got[1], again[1]))
        assert all(torch.equal(a, b) for a, b in zip(got[0], only_rec[0]))  # records only: same records, no ops
        assert all(o.numel() == 0 for o in only_rec[1])
    else:
        assert only_rec is None
    if rank == 0:
        recs, opss = got
        scores, cigars = [], []
        for rr, oo in zip(recs, opss):
            rr, oo = rr.numpy(), oo.numpy().view(np.uint64)
            for i in range(rr.shape[0]):
                scores.append(int(rr[i, 1]))
                cigars.append(O.ops_to_cigar(oo[int(rr[i, 11]):int(rr[i, 11]) + int(rr[i, 10])]))
        np.save(out_path, np.array([scores, cigars], dtype=object), allow_pickle=True)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    from wfa_amd.shard import shard_range
    for n in (0, 1, 7, 8, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gather_matches_single_process(tmp_path, built):
    sys.path.insert(0, ROOT)
    import wfa_amd as w
    from oracle import oracle as O
    n_total = 301  # odd: the two shards differ in size
    out = str(tmp_path / "gathered.npy")
    mp.start_processes(_worker, args=(2, _free_port(), n_total, out), nprocs=2, join=True, start_method="spawn")
    scores, cigars = np.load(out, allow_pickle=True)
    data = w.generate_pairs(seed=9, n_pairs=n_total, length=120, error_rate=0.05, n_threads=1)
    ref = O.align_batch(O.make_params(adaptive=(10, 50, 1)), *data)
    assert list(scores) == [int(x) for x in ref.score]
    assert list(cigars) == [ref.cigar(i) for i in range(n_total)]
