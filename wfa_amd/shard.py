"""Pair sharding and result gather for multi-GPU runs (one process per GPU, torch.distributed).

Alignments share no state (the reference's own model is one Aligner per goroutine, wfa.go:73-78), so a batch
is cut into contiguous shards of pairs, one per rank, with NO collective on the data path.  The only exchange
is the gather of results onto rank 0: fixed-size 64-byte records and -- optionally -- the CIGAR op arrays padded to
the largest shard (counts are agreed with one tiny all-gather).  Backend "nccl" is RCCL over xGMI on the GPU box;
the same code runs over "gloo" on CPU tensors in the tests.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[begin, end) of the pairs rank `rank` owns: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class PendingGather:
    """A result gather in flight (see gather_results_async).  `wait()` completes it: on a CUDA/HIP device the
    current stream is made to wait for the collective, not the host."""

    def __init__(self, works, keep, recs, opss, all_sizes, is_dst):
        self._works, self._keep = works, keep  # `keep` pins the send buffers until the collective is done
        self._recs, self._opss, self._sizes, self._is_dst = recs, opss, all_sizes, is_dst

    def wait(self) -> Optional[Tuple[List[torch.Tensor], List[torch.Tensor]]]:
        for wk in self._works:
            wk.wait()
        self._works, self._keep = [], None
        if not self._is_dst:
            return None
        return ([r[:int(s[0])] for r, s in zip(self._recs, self._sizes)],
                [o[:int(s[1])] for o, s in zip(self._opss, self._sizes)])


def gather_results_async(rec: torch.Tensor, ops: torch.Tensor, n_ops: int, dst: int = 0,
                         with_ops: bool = True) -> PendingGather:
    """Start the gather of every rank's result records [n_i, 16] (int32) and the used prefix of its op buffer
    (int64) onto `dst` and return at once.  The send buffers are private copies, so the caller may overwrite
    `rec` / `ops` (the next batch) while the exchange runs on the collective's own stream.

    Shards may differ in size, so both arrays are padded to the largest shard (sizes are agreed with one tiny
    all-gather).  `wait()` returns (records per rank, ops per rank) on `dst`, trimmed back to each rank's true
    sizes, None elsewhere.  The OPS_OFF fields of a record index into that rank's own op array.

    with_ops = False gathers the records only (score, region, statistics, op counts: 64 bytes per pair); the CIGAR
    op arrays -- twelve times the volume at 1 kbp -- stay in the HBM of the GPU that produced them, like the results
    of a single-GPU run, and the per-rank op lists come back empty.
    """
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = rec.device
    sizes = torch.tensor([rec.shape[0], n_ops], dtype=torch.int64, device=dev)
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    max_rec = max(int(s[0]) for s in all_sizes)
    max_ops = max(int(s[1]) for s in all_sizes)

    def padded_copy(t: torch.Tensor, rows: int) -> torch.Tensor:
        out = torch.empty((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        out[:t.shape[0]] = t
        if rows > t.shape[0]:
            out[t.shape[0]:] = 0
        return out

    rec_p = padded_copy(rec, max_rec)
    recs  = [torch.empty_like(rec_p) for _ in range(world)] if rank == dst else None
    w1    = dist.gather(rec_p, recs, dst=dst, async_op=True)
    if not with_ops:
        opss = [ops[:0] for _ in range(world)] if rank == dst else None
        for s in all_sizes:
            s[1] = 0
        return PendingGather([w1], (rec_p,), recs, opss, all_sizes, rank == dst)
    ops_p = padded_copy(ops[:n_ops], max_ops)
    opss  = [torch.empty_like(ops_p) for _ in range(world)] if rank == dst else None
    w2    = dist.gather(ops_p, opss, dst=dst, async_op=True)
    return PendingGather([w1, w2], (rec_p, ops_p), recs, opss, all_sizes, rank == dst)


def gather_results(rec: torch.Tensor, ops: torch.Tensor, n_ops: int, dst: int = 0, with_ops: bool = True
                   ) -> Optional[Tuple[List[torch.Tensor], List[torch.Tensor]]]:
    """Blocking form of gather_results_async."""
    return gather_results_async(rec, ops, n_ops, dst, with_ops).wait()
