"""Child process of tests/test_scale_gpu.py::test_one_rank_rccl_gather_against_oracle: ONE rank with backend `nccl`
(= RCCL on ROCm) on GPU 0.  It aligns 2 000 pairs through the C-ABI into torch tensors in HBM, runs the collectives of the
multi-GPU path on those DEVICE tensors -- wfa_amd/shard.py:gather_results_async with and without the CIGAR op arrays, plus
the barrier / max-reduction / rank census bench.py makes -- and writes what rank 0 received to the .npz named on the
command line.  The parent compares that with the oracle.  (The process group is created before anything else touches the
GPU, and nothing here re-executes the process.)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, n_pairs, seed):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    import wfa_amd as w
    from wfa_amd import _lib as L
    from wfa_amd.shard import gather_results_async
    blob, q_off, q_len, t_off, t_len = w.generate_pairs(seed=seed, n_pairs=n_pairs, length=1000, error_rate=0.05)
    d = [torch.from_numpy(a).to(dev) for a in (blob, q_off.view(np.int64), q_len.view(np.int32), t_off.view(np.int64), t_len.view(np.int32))]
    ops_cap = int(q_len.sum() + t_len.sum()) // 4 + 8 * n_pairs + 1024
    d_rec = torch.zeros((n_pairs, L.REC_WORDS), dtype=torch.int32, device=dev)
    d_ops = torch.zeros(ops_cap, dtype=torch.int64, device=dev)
    al = w.New(w.DefaultPenalties, w.Options(GlobalAlignment=True), device=0)
    assert al.AdaptiveReduction(w.DefaultAdaptiveOption) is None
    prm, needed = al._params(), C.c_uint64()
    L.check(L.lib().wfahip_align_batch_device(al._ctx, C.byref(prm), d[0].data_ptr(), blob.size, d[1].data_ptr(), d[2].data_ptr(),
                                              d[3].data_ptr(), d[4].data_ptr(), n_pairs, 0, d_rec.data_ptr(), d_ops.data_ptr(), ops_cap,
                                              C.byref(needed), torch.cuda.current_stream(dev).cuda_stream), "wfahip_align_batch_device")
    n_ops = int(needed.value)
    dist.barrier()
    # records only (what bench.py gathers every step), then records + the used prefix of the op buffer
    pend = gather_results_async(d_rec[:, :L.REC_OPS_OFF_LO], d_ops, n_ops, dst=0, with_ops=False)
    d_rec_copy = d_rec.clone()
    got1 = pend.wait()
    got2 = gather_results_async(d_rec_copy, d_ops, n_ops, dst=0, with_ops=True).wait()
    tt = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([n_pairs], dtype=torch.int64, device=dev)
    cnts = [torch.empty_like(cnt)]
    dist.all_gather(cnts, cnt)
    torch.cuda.synchronize(dev)
    assert got1[0][0].is_cuda and got2[1][0].is_cuda  # the exchange ran on HBM tensors
    np.savez(out_path, rec_only=got1[0][0].cpu().numpy(), rec_full=got2[0][0].cpu().numpy(), ops=got2[1][0].cpu().numpy(),
             n_ops=np.int64(n_ops), reduced=tt.cpu().numpy(), census=cnts[0].cpu().numpy(), ops_only_len=np.int64(got1[1][0].numel()))
    w.RecycleAligner(al)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
