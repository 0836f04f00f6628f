#!/usr/bin/env python3
"""bench.py -- aligned pairs/s of the HIP wavefront-alignment hot path on MI355X.

Default workload (BASELINE.json configs[2], the one `metric` is quoted on; --config c3): 1e6 synthetic 1 kbp DNA
pairs at 5 % error per GPU, global alignment, wf-adaptive 10/50/1, penalties 4/6/2, seed 3.  A "step" is one pass of
the hot path over the rank's batch: raw byte sequences already resident in HBM -> result records + CIGAR ops in HBM.

Other workloads of BASELINE.json, each printing its own JSON line:
  --config c2    configs[1]: 1e5 x 150 bp @2 %, global, wf-adaptive off, seed 2   (c2m: the same pairs, 1e6 of them)
  --config c4    configs[3]: 1e7 x 1 kbp @5 % IN TOTAL (strong scaling: --total-pairs 10000000), seed 4
  --config c5s   configs[4] sample: 8 x 100 kbp @10 %, semi-global, wf-adaptive 10/50/1, seed 5
  --config k10 | k20 | l5 | l10 | l20   the reference's published grid (1e5 x 1 kbp, 500 x 50 kbp); L5: 2e4 x 50 kbp

Multi-GPU: one rank per GPU, pairs sharded over ranks, no data-path collective; the only RCCL traffic is the gather of
the result records (44 bytes per pair) onto rank 0 after every step (--gather-ops ships the CIGAR op arrays too).
`--gpus N` with WORLD_SIZE unset starts the N ranks itself (a torch.distributed.run child process, started before
this process touches a GPU); under torchrun (WORLD_SIZE set) the process is one of the ranks.  Weak scaling by
default (--pairs per GPU); --total-pairs T shards T pairs over the ranks (strong scaling).

--backend gloo --dry runs the same multi-rank control flow on CPU tensors without the kernels (tests/test_bench_gloo.py).

Prints ONE JSON line (rank 0).  `value` = pairs aligned by all ranks / max-over-ranks wall time of K steps.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "aligned pairs/sec (and Gcells/s) on 1e6 synthetic 1 kbp pairs @5% error"

CONFIGS = {
    # name: (pairs, length, error, seed, semi_global, adaptive, total_pairs, cpu_sample)
    "c3": dict(pairs=1_000_000, length=1000, error=0.05, seed=3, semi_global=False, adaptive=True, total=0, cpu=150_000),
    "c2": dict(pairs=100_000, length=150, error=0.02, seed=2, semi_global=False, adaptive=False, total=0, cpu=100_000),
    # configs[1]'s pairs in a number that fills the GPU (1e5 pairs are less than one pair per lane slot of the chip)
    "c2m": dict(pairs=1_000_000, length=150, error=0.02, seed=2, semi_global=False, adaptive=False, total=0, cpu=300_000),
    "c4": dict(pairs=0, length=1000, error=0.05, seed=4, semi_global=False, adaptive=True, total=10_000_000, cpu=150_000),
    "c5s": dict(pairs=8, length=100_000, error=0.10, seed=5, semi_global=True, adaptive=True, total=0, cpu=8),
    # the reference's own published grid beyond the headline row (README.md:326-345 = benchmark.tsv:4-19: wfa-go -N -i,
    # global, wf-adaptive 10/50/1, 1e5 x 1 kbp and 500 x 50 kbp at 5 / 10 / 20 % error; 1 kbp @5 % is c3 at 1e6 pairs)
    "k10": dict(pairs=100_000, length=1000, error=0.10, seed=10, semi_global=False, adaptive=True, total=0, cpu=60_000),
    "k20": dict(pairs=100_000, length=1000, error=0.20, seed=20, semi_global=False, adaptive=True, total=0, cpu=25_000),
    "l5": dict(pairs=500, length=50_000, error=0.05, seed=55, semi_global=False, adaptive=True, total=0, cpu=500),
    # ... and the same pairs in a number that fills the GPU (2e4 x 50 kbp: 5 000 waves of four pairs)
    "L5": dict(pairs=20_000, length=50_000, error=0.05, seed=55, semi_global=False, adaptive=True, total=0, cpu=200),
    "l10": dict(pairs=500, length=50_000, error=0.10, seed=510, semi_global=False, adaptive=True, total=0, cpu=300),
    "l20": dict(pairs=500, length=50_000, error=0.20, seed=520, semi_global=False, adaptive=True, total=0, cpu=120),
    # round 5, off the 2 : 4 / global rails: the headline's pairs under penalties of another shape (2/4/2: x : o+e : e = 1 : 3 : 1,
    # wfa.go:32-36 takes any), and the headline's pairs aligned semi-global (wfa-go -g, wfa-go/wfa-go.go:96: seeds on the whole
    # first row and column, wfa.go:163-183, until wf-adaptive has collapsed the band)
    "p242": dict(pairs=1_000_000, length=1000, error=0.05, seed=3, semi_global=False, adaptive=True, total=0, cpu=150_000, pen=(2, 4, 2)),
    "g3": dict(pairs=1_000_000, length=1000, error=0.05, seed=3, semi_global=True, adaptive=True, total=0, cpu=20_000),
    # configs[4] sample at 32 pairs: every team draws wide pairs
    "c5s32": dict(pairs=32, length=100_000, error=0.10, seed=5, semi_global=True, adaptive=True, total=0, cpu=0),
}
# steps of the default run: timed regions of a few seconds
DEFAULT_STEPS = {"c3": 250, "c2": 10000, "c2m": 2000, "c4": 25, "c5s": 4, "k10": 400, "k20": 150, "l5": 150, "L5": 20, "l10": 60, "l20": 25,
                 "p242": 200, "g3": 5, "c5s32": 1}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: enough for ~5 s on the default workload)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--pairs", type=int, default=None, help="pairs per GPU (weak scaling)")
    ap.add_argument("--total-pairs", type=int, default=None, help="pairs over all GPUs (strong scaling)")
    ap.add_argument("--length", type=int, default=None)
    ap.add_argument("--error", type=float, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--semi-global", action="store_true", default=None)
    ap.add_argument("--no-adaptive", action="store_true", default=None)
    ap.add_argument("--penalties", default=None, help="mismatch,gap_open,gap_ext (default: the configuration's, 4,6,2)")
    ap.add_argument("--cpu-sample", type=int, default=None, help="pairs timed on host cores (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=None, help="threads of the cpu_baseline leg (default 1; c5s: 8, one pair each)")
    ap.add_argument("--cpu-all-cores", type=int, default=1, help="1: also time the oracle on every host core (N = 1 only)")
    ap.add_argument("--host-entry", type=int, default=1, help="1: also time wfahip_align_batch (host blobs -> host results), N = 1 only")
    ap.add_argument("--latency", type=int, default=1, help="1: also time single-pair Align round trips, N = 1 only")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (experiments)")
    ap.add_argument("--gather-ops", action="store_true", help="multi-GPU: gather the CIGAR op arrays onto rank 0 too")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl")
    ap.add_argument("--dry", action="store_true", help="no kernels, CPU tensors: exercises the multi-rank control flow only")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--share-gpus", action="store_true",
                    help="N ranks on fewer GPUs (rank r uses GPU r mod device_count): every rank runs the real kernels on "
                         "its own shard, the record gather goes over gloo through host copies (two RCCL ranks cannot share "
                         "a device).  For exercising the sharded path on a 1-GPU box; not a scaling measurement")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: create the RCCL process group anyway (world size 1) and run every collective of the multi-GPU "
                         "path -- barrier, the asynchronous record gather, the max-over-ranks reduction, the rank census -- on "
                         "device tensors, so the `nccl` lines of this file execute on a 1-GPU box")
    ap.add_argument("--other-configs", type=int, default=None,
                    help="1: after the timed region of the default (c3, N = 1) run, short legs of c2, k10, l5 and c5s on the same "
                         "GPU, reported under config.other_configs (default: 1 for the default run, else 0)")
    ap.add_argument("--plan", action="store_true",
                    help="no GPU, no processes: print ONE JSON line with what each of the --gpus ranks would hold in HBM for this "
                         "configuration (input blob, records, CIGAR ops, wavefront arenas per chunk) -- an out-of-memory condition shows "
                         "before a multi-GPU run is launched")
    ap.add_argument("--hbm-gib", type=float, default=288.0, help="--plan: HBM per GPU (MI355X: 288)")
    ap.add_argument("--dump-records", default=None,
                    help="rank 0 writes the records gathered in the last timed step (all ranks, pair order) to this .npy file")
    args = ap.parse_args(argv)
    c = CONFIGS[args.config]
    if args.pairs is None:
        args.pairs = c["pairs"]
    if args.total_pairs is None:
        args.total_pairs = c["total"]
    if args.length is None:
        args.length = c["length"]
    if args.error is None:
        args.error = c["error"]
    if args.seed is None:
        args.seed = c["seed"]
    if args.semi_global is None:
        args.semi_global = c["semi_global"]
    if args.no_adaptive is None:
        args.no_adaptive = not c["adaptive"]
    args.pen = tuple(int(v) for v in args.penalties.split(",")) if args.penalties else tuple(c.get("pen", (4, 6, 2)))
    if args.cpu_sample is None:
        args.cpu_sample = c["cpu"]
    if args.cpu_threads is None:
        args.cpu_threads = 8 if args.config in ("c5s", "c5s32") else 1  # (a 100 kbp semi-global pair is minutes of one core)
    if args.steps is None:
        args.steps = DEFAULT_STEPS[args.config]  # (timed regions of ~5 s; c5s ~3 s)
    if args.warmup is None:
        args.warmup = 1 if args.config in ("c5s", "c5s32", "g3") else 3
    if args.other_configs is None:
        plain = (args.config == "c3" and args.gpus == 1 and not args.dry and not args.opt and args.pairs == c["pairs"] and
                 args.total_pairs == c["total"] and args.length == c["length"] and args.error == c["error"])
        args.other_configs = 1 if plain else 0
    return args


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv):
    """--gpus N without a launcher: start the N ranks as children of a torch.distributed.run process.  Nothing in this
    process has touched a GPU yet (no torch.cuda / HIP call), and it is never re-exec'd: it waits for the child and
    leaves with its return code."""
    port = args.master_port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def plan(args):
    """What every rank of `--gpus N` holds in HBM (no GPU is touched): the shard's input, its result buffers and the arenas of
    the first-pass kernel as the library sizes them (wfa_amd/csrc/wfa_host.hip: a chunk of pairs owns its arena slots until
    its backtrace has run; chunks take at most 35 % of HBM; long-pair ladders at most 60 %)."""
    import math
    sys.path.insert(0, ROOT)
    from wfa_amd.shard import shard_range
    world, hbm = max(1, args.gpus), args.hbm_gib * 2 ** 30
    L, e = args.length, args.error
    stride = (L + 15) // 16 * 16 + (int(L * (1 + e)) + 16 + 15) // 16 * 16  # (upper bound of the generator's bytes per pair)
    ranks = []
    for r in range(world):
        first, end = shard_range(args.total_pairs, r, world) if args.total_pairs > 0 else (r * args.pairs, (r + 1) * args.pairs)
        n = end - first
        blob = n * stride + 16
        idx = n * (8 + 8 + 4 + 4)
        rec = n * 64
        sum_len = n * (2 * L + int(L * e))
        ops = 8 * ((sum_len + 2 * n + 1024) if (args.semi_global or L >= 20000) else int(sum_len * max(0.25, 3.0 * e)) + 8 * n + 1024)
        if L <= 240:
            slot = 4 * 8 * L          # rows of 32 x 16 bit, 8 L words per pair
        elif L <= 2000 and not args.semi_global:
            slot = 4 * 8 * L          # 16-bit tiles of wfa_duo_kernel
        elif not args.semi_global:
            slot = 4 * 16 * L         # 32-bit tiles / plain rows of the sliding-window instances
        else:
            slot = 0                  # long-pair ladder: one pool
        if slot:
            chunk = max(1, min(n, int(hbm * 0.35) // slot))
            arena, n_chunks = chunk * slot, math.ceil(n / chunk)
            prepack = chunk * 4 * (4 + 2 * ((L + int(L * e) + 16 + 15) // 16 + 2))
        else:
            arena, n_chunks, prepack = int(hbm * 0.60), 1, 0
        total = blob + idx + rec + ops + arena + prepack + n * 16 + n * 8
        ranks.append({"rank": r, "pairs": n, "first_pair": first, "input_gib": (blob + idx) / 2 ** 30, "records_gib": rec / 2 ** 30,
                      "ops_gib": ops / 2 ** 30, "arena_gib": arena / 2 ** 30, "chunks": n_chunks, "prepack_gib": prepack / 2 ** 30,
                      "total_gib": total / 2 ** 30, "fits": total <= 0.95 * hbm})
    print(json.dumps({"plan": args.config, "n_gpus": world, "hbm_gib": args.hbm_gib, "pairs_total": sum(r["pairs"] for r in ranks),
                      "gather_bytes_per_rank_step": [r["pairs"] * 44 for r in ranks], "all_fit": all(r["fits"] for r in ranks), "ranks": ranks}), flush=True)
    return 0 if all(r["fits"] for r in ranks) else 3


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.plan:
        sys.exit(plan(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, argv))
    run_rank(args)


def run_rank(args):
    import ctypes as C

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dry = args.dry
    share = bool(args.share_gpus) and not dry
    if dry:
        dev = torch.device("cpu")
        dev_index = 0
    else:
        # (device_count() does not initialise the GPU on this image; --share-gpus: rank r on GPU r mod device_count)
        dev_index = local_rank % max(1, torch.cuda.device_count()) if share else local_rank
        dev = torch.device(f"cuda:{dev_index}")
    backend = "gloo" if (dry or share or args.backend == "gloo") else "nccl"
    # tensors handed to collectives: HBM for RCCL, host copies for gloo (--share-gpus: the records are staged through
    # the host, because two RCCL ranks cannot sit on one device)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    # collectives run with more than one rank -- or with ONE when --force-collective asks for the process group anyway
    coll = world > 1 or bool(args.force_collective)
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(args.master_port or free_port()))
            os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    if not dry:
        torch.cuda.set_device(dev_index)

    import __graft_entry__ as entry
    if rank == 0:  # one rank builds (a no-op when the in-tree library is current), the others wait for it
        entry.build()
    if coll:
        dist.barrier()
    if rank != 0:
        entry.build()
    import wfa_amd as w
    from wfa_amd import _lib as L
    from wfa_amd.shard import gather_results_async, shard_range

    # ---- this rank's shard
    if args.total_pairs > 0:
        first, end = shard_range(args.total_pairs, rank, world)
        scaling = "strong"
    else:
        first, end = rank * args.pairs, (rank + 1) * args.pairs
        scaling = "weak"
    n = end - first
    n_all = args.total_pairs if args.total_pairs > 0 else args.pairs * world

    # ---- synthetic input (host generation, then H2D: outside the timed region)
    blob, q_off, q_len, t_off, t_len = w.generate_pairs(args.seed, n, args.length, args.error, first_index=first,
                                                        n_threads=min(32, os.cpu_count() or 8))
    max_len = int(max(q_len.max(), t_len.max())) if n else 1
    sum_len = int(q_len.astype(np.int64).sum() + t_len.astype(np.int64).sum())
    # (the backtrace reserves score / min(x, e) * 2 + 8 op slots per pair before it knows the CIGAR: ~2.5 x error rate x
    # the pair's bases at 4/6/2)
    ops_cap = int(sum_len * max(0.25, 3.0 * args.error)) + 8 * n + 1024
    if args.semi_global or args.length >= 20000:
        ops_cap = sum_len + 2 * n + 1024
    d_rec = torch.zeros((max(n, 1), L.REC_WORDS), dtype=torch.int32, device=dev)[:n]
    d_ops = torch.zeros(ops_cap if not dry else 64 * n + 16, dtype=torch.int64, device=dev)

    timing = L.Timing()
    al = None
    if not dry:
        d_blob = torch.from_numpy(blob).to(dev)
        d_qoff = torch.from_numpy(q_off.view(np.int64)).to(dev)
        d_toff = torch.from_numpy(t_off.view(np.int64)).to(dev)
        d_qlen = torch.from_numpy(q_len.view(np.int32)).to(dev)
        d_tlen = torch.from_numpy(t_len.view(np.int32)).to(dev)
        al = w.New(w.Penalties(*args.pen), w.Options(GlobalAlignment=not args.semi_global), device=dev_index)
        if not args.no_adaptive:
            assert al.AdaptiveReduction(w.DefaultAdaptiveOption) is None
        if args.opt:
            os.environ.setdefault("WFAHIP_DEBUG", "1")  # (--opt sets experiment knobs: refused otherwise)
        for kv in args.opt:
            key, val = kv.split("=")
            L.check(L.lib().wfahip_set_option(al._ctx, key.encode(), int(val)), "wfahip_set_option")
        prm = al._params()
        lib = L.lib()
        # (the device pointers of the resident batch: looked up once, not in every step -- the call is the C-ABI's)
        dev_ptrs = (d_blob.data_ptr(), blob.size, d_qoff.data_ptr(), d_qlen.data_ptr(), d_toff.data_ptr(), d_tlen.data_ptr(),
                    d_rec.data_ptr(), d_ops.data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
    pending = [None]  # the result gather in flight (multi-GPU)
    gather_bytes = [0]

    def step():
        if dry:
            # stand-in for the kernels: a deterministic record per pair (score field = global pair index)
            d_rec[:, L.REC_STATUS] = 0
            d_rec[:, L.REC_SCORE] = torch.arange(first, end, dtype=torch.int32)
            d_rec[:, L.REC_OPS_LEN] = 1
            n_ops = n
        else:
            needed = C.c_uint64()
            rc = lib.wfahip_align_batch_device(al._ctx, C.byref(prm), *dev_ptrs[:2], *dev_ptrs[2:6], n, max_len,
                                               *dev_ptrs[6:], ops_cap, C.byref(needed), stream)
            L.check(rc, "wfahip_align_batch_device")
            lib.wfahip_last_timing(al._ctx, C.byref(timing))
            n_ops = int(needed.value)
        if coll:
            # Result gather onto rank 0 over RCCL/xGMI (fixed-size records; with --gather-ops the padded op arrays
            # too).  It is started here and completed before the next one starts (or at the end of the timed
            # region), so the exchange of batch i runs beside the alignment of batch i+1; the send buffers are
            # private copies.  (records only: the first 11 words -- status, score, region, statistics, op count; the
            # op offsets, cell census and score count that follow only mean something next to the rank's own ops)
            if pending[0] is not None:
                pending[0].wait()
            rec_out = d_rec if args.gather_ops else d_rec[:, :L.REC_OPS_OFF_LO]
            gather_bytes[0] = rec_out.shape[0] * rec_out.shape[1] * 4 + (8 * n_ops if args.gather_ops else 0)
            ops_out = d_ops
            if cdev != dev:  # gloo beside real kernels: stage through the host (the D2H copy waits for the kernels)
                rec_out = rec_out.to(cdev)
                ops_out = d_ops[:n_ops].to(cdev) if args.gather_ops else d_ops[:0].to(cdev)
            pending[0] = gather_results_async(rec_out, ops_out, n_ops, dst=0, with_ops=args.gather_ops)
        return n_ops

    last_gather = [None]

    def drain():
        if pending[0] is not None:
            last_gather[0] = pending[0].wait()
            pending[0] = None

    def sync_all():
        if coll:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize(dev)

    # Setup, outside the timed region like the input upload: the first call of a context allocates its device buffers
    # (61 GiB of wavefront arenas for the default workload: seconds of hipMalloc).  With --warmup >= 1 the warm-up
    # steps do that; with --warmup 0 one extra untimed call does, reported as config.setup_steps.
    setup_steps = 1 if args.warmup == 0 else 0
    for _ in range(args.warmup + setup_steps):
        step()
    drain()
    kernel_ms, main_ms = [], []
    n_ops = 0
    sync_all()
    smi_before = gpu_telemetry(dev_index) if (rank == 0 and not dry) else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_ops = step()
        kernel_ms.append(timing.kernel_ms)
        main_ms.append(timing.main_kernel_ms)
    drain()  # the last gather completes inside the timed region
    sync_all()
    elapsed = time.perf_counter() - t0
    # what the run ran AT (round 5: the round-4 driver measured the unchanged headline kernel 7 % slower than the builder's box and
    # nothing in the line could say why): rocm-smi's clocks / power / temperature right before and right after the timed region,
    # and the shader clock under load by the library's own probe (s_memtime against the 100 MHz s_memrealtime on every SIMD)
    smi_after = gpu_telemetry(dev_index) if (rank == 0 and not dry) else None
    clock = None
    if rank == 0 and not dry:
        f0, f1, f2 = C.c_double(), C.c_double(), C.c_double()
        if lib.wfahip_debug_clock(al._ctx, C.byref(f0), C.byref(f1), C.byref(f2)) == 0:
            clock = {"mean": f0.value, "min": f1.value, "max": f2.value}
    tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if coll:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())

    # one more gather on its own, timed (how long the exchange takes when nothing hides it)
    gather_ms = None
    if coll:
        sync_all()
        t1 = time.perf_counter()
        rec_out = d_rec if args.gather_ops else d_rec[:, :L.REC_OPS_OFF_LO]
        ops_out = d_ops
        if cdev != dev:
            rec_out = rec_out.to(cdev)
            ops_out = d_ops[:n_ops].to(cdev) if args.gather_ops else d_ops[:0].to(cdev)
        gather_results_async(rec_out, ops_out, n_ops, dst=0, with_ops=args.gather_ops).wait()
        sync_all()
        gather_ms = (time.perf_counter() - t1) * 1e3
        # per-rank pair counts as the collective saw them
        cnt = torch.tensor([n], dtype=torch.int64, device=cdev)
        cnts = [torch.empty_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        pairs_per_rank = [int(c.item()) for c in cnts]
        n_seen = dist.get_world_size()
        gathered_ok = None
        if rank == 0 and last_gather[0] is not None:
            recs = last_gather[0][0]
            gathered_ok = [int(r.shape[0]) for r in recs] == pairs_per_rank
            if dry:  # the records of every rank arrived, in rank order, untouched
                scores = torch.cat([r[:, L.REC_SCORE] for r in recs]).cpu()
                gathered_ok = gathered_ok and bool(torch.equal(scores, torch.arange(n_all, dtype=torch.int32)))
            if args.dump_records:  # (tests: the gathered records of ALL shards against the oracle, in pair order)
                np.save(args.dump_records, torch.cat([r for r in recs]).cpu().numpy())
    else:
        pairs_per_rank, n_seen, gathered_ok = [n], 1, None
        if args.dump_records and rank == 0:
            np.save(args.dump_records, d_rec[:, :L.REC_OPS_OFF_LO].cpu().numpy())

    # ---- accounting for the roofline (algorithmic bytes): one more pass, untimed, with the forward
    # kernel's count of stored wavefront words switched on (instrumentation: off in the timed steps, where it would
    # cost 4 % of the forward pass; the dataset is the same, so the count is that of every timed step)
    if not dry:
        L.check(lib.wfahip_set_option(al._ctx, b"census", 1), "census")
        step()
        drain()
        sync_all()
        L.check(lib.wfahip_set_option(al._ctx, b"census", 0), "census")
    rec = d_rec.cpu().numpy().view(np.uint32)
    ok = rec[:, L.REC_STATUS] == 0
    cells = int(rec[:, L.REC_CELLS_LO].astype(np.uint64).sum()) + (int(rec[:, L.REC_CELLS_HI].astype(np.uint64).sum()) << 32)
    n_ops_total = int(rec[:, L.REC_OPS_LEN].astype(np.uint64).sum())
    alg_bytes = 4 * cells + sum_len + 64 * n + 8 * n_ops_total
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else 0.0
    main_k_ms = float(np.mean(main_ms)) if main_ms else 0.0
    achieved = alg_bytes / (main_k_ms * 1e-3) / 1e9 if main_k_ms > 0 else 0.0  # GB/s over the dominant kernel's launches of one step

    kname = KNAMES[min(int(timing.main_kernel_kind), len(KNAMES) - 1)]
    if kname == "wfa_duo_kernel":  # (round 5: an instance per penalty shape -- x/g and (o+e)/g are its template arguments; <false: without the census)
        import math
        g = math.gcd(math.gcd(args.pen[0], args.pen[1] + args.pen[2]), args.pen[2])
        kname = f"wfa_duo_kernel<false, {args.pen[0] // g}, {(args.pen[1] + args.pen[2]) // g}>"

    out = None
    if rank == 0:
        # HBM traffic + instruction counts of the dominant kernel from the committed rocprofv3 PMC passes of this
        # same command (profiles/*_pmc.json, made by scripts/profile_bench.sh; PMC cannot be collected from inside the
        # bench).  gfx950: FETCH_SIZE counts half the bytes of the coalesced input reads (checked against the known
        # 2.0 GB of sequence bytes the forward kernel must read: it reports 1.12 GB), so it is doubled; WRITE_SIZE is
        # taken as is.  `stale` says whether the profile was taken on another build of the kernels.
        pm = find_profile(args.config, kname) if (not dry and n == CONFIGS[args.config]["pairs"] and not args.opt) else None
        traffic = pm["traffic"] if pm else None
        total_pairs = n_all * args.steps
        value = total_pairs / elapsed
        cfg = {"workload": f"{n_all if scaling == 'strong' else n} x {args.length} bp pairs{'' if scaling == 'strong' else '/GPU'} "
                           f"@{args.error:.0%} error, {'semi-global' if args.semi_global else 'global'} gap-affine {'/'.join(str(v) for v in args.pen)}, "
                           f"wf-adaptive {'off' if args.no_adaptive else '10/50/1'}, seed {args.seed}",
               "config": args.config, "pairs_per_gpu": n, "pairs_per_rank": pairs_per_rank, "total_pairs_per_step": n_all,
               "length": args.length, "error_rate": args.error,
               "parallelism": f"pair-sharded x{world}", "ranks_seen_by_collective": n_seen, "backend": backend if coll else "none (1 GPU)",
               "collective_forced": bool(coll and world == 1),
               "gpus_shared": (f"{world} ranks on {torch.cuda.device_count()} GPU(s): sharded path exercised, NOT a scaling measurement"
                               if share and world > 1 else None),
               "status_ok": int(ok.sum()), "setup_steps": setup_steps,
               "gather": ("records + CIGAR ops" if args.gather_ops else "records") if coll else "none (1 GPU)",
               "gather_bytes_per_rank_step": gather_bytes[0] if coll else 0, "gather_ms_standalone": gather_ms,
               "gathered_records_complete": gathered_ok,
               "gcells_per_s": value * args.length * args.length / 1e9,
               "kernel_ms_per_step": k_ms, "main_kernel_ms": main_k_ms, "launches_per_step": int(timing.n_launches),
               "packed_pairs": int(timing.n_packed_pairs),
               "retried_pairs": int(timing.n_retried_pairs), "arena_gib": timing.arena_bytes / 2 ** 30,
               "wf_cells_per_pair": cells / max(n, 1), "cigar_ops_per_pair": n_ops_total / max(n, 1),
               "timed_region_s": elapsed,
               "clock_mhz_under_load": clock["mean"] if clock else None, "clock_mhz_under_load_range": [clock["min"], clock["max"]] if clock else None,
               "clock_note": "shader MHz while the library's probe kernel keeps every SIMD issuing (s_memtime / s_memrealtime), right after the timed region",
               "gpu_sysfs_before": smi_before, "gpu_sysfs_after": smi_after,
               "steady_state": f"the timed steps re-align the resident batch after {args.warmup + setup_steps} untimed call(s) of the same workload class: "
                               "what the context learns per class (rows per pair, first window, arena level) is in place; a first call "
                               "of a class also allocates its arenas (seconds, include/wfa_hip.h)"}
        if dry:
            cfg["dry"] = True
        metric = METRIC if args.config in ("c3", "c4") else (
            f"aligned pairs/sec (and Gcells/s) on {n} synthetic {args.length} bp pairs @{args.error:.0%} error")
        out = {"metric": metric, "value": value, "unit": "pairs/s", "n_gpus": n_seen, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
               "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": cfg}
        if not dry:
            roof = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                    "frac": achieved / 8000.0, "traffic": traffic,
                    "traffic_source": pm["source"] if pm else None, "traffic_stale": pm["stale"] if pm else None,
                    "frac_on_traffic": (traffic / (main_k_ms * 1e-3) / 8e12) if (traffic and main_k_ms > 0) else None,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "kernel": (pm["kernel"] if pm else None) or (kname + (", ..>" if kname.startswith("wfa_blk_kernel") and not kname.endswith(">") else "")),
                    "kernel_ms": main_k_ms, "all_kernels_ms": k_ms,
                    "note": "achieved = algorithmic bytes of one step / duration of the dominant kernel's launches in "
                            "that step (HIP events on the launch stream); peak = 8 TB/s HBM3E spec; the forward kernel is "
                            "integer-VALU-issue bound, not HBM bound: see `secondary` (DESIGN.md section 5)"}
            if pm and pm.get("valu_insts") and main_k_ms > 0:
                # second ceiling: vector-instruction issue.  `peak` is the guide's figure: 1024 SIMDs x 2.4 GHz / 2 cycles per
                # wave64 instruction (MI355X_MICROARCH.md).  `attainable` is what a SIMD was MEASURED to issue for a stream that
                # mixes the forms these kernels are made of (max / cmp / cndmask / DPP with adds) at the kernel's own occupancy,
                # in nanoseconds per SIMD instruction so that no clock is assumed (profiles/r03_valu_issue_probe.txt, the
                # calibrated probe: waves per SIMD asserted from HW_ID, ticks checked against s_memrealtime and HIP events):
                # 1.333 ns at 4 waves per SIMD (wfa_duo_kernel), 1.079 ns at 5 (wfa_blk_kernel<16,1>), 1.080 ns at 8.
                peak = 256 * 4 * 2.4e9 / 2 / 1e9
                ach = pm["valu_insts"] / (main_k_ms * 1e-3) / 1e9
                # (wfa_lane_kernel: LDS allows two waves per SIMD -- 1.799 ns, the same row of the probe at that occupancy)
                # (the wave-per-pair instances of small long-read batches, wfa_blk_kernel<64, ..>: ONE wave on each SIMD they use -- 1.976 ns
                # per instruction, the probe's lone-wave row -- and as many SIMDs as there are pairs)
                # (by the kernel KIND, not the name: kinds 5 and 13 are wfa_blk_kernel<64, ..> too, but with four diagonals per lane and up to
                # sixteen waves per CU on a GPU-filling pass)
                lone = int(timing.main_kernel_kind) in (14, 15, 16)
                waves = 1 if lone else 2 if "lane" in kname else 4 if "duo" in kname else (5 if kname.startswith("wfa_blk_kernel<16") else 4)
                ns_per = {1: 1.976, 2: 1.799, 4: 1.333, 5: 1.079, 8: 1.080}[waves]
                attainable = (min(n, 256 * 4) if lone else 256 * 4) / ns_per
                roof["secondary"] = {"bound": "valu-issue", "achieved": ach, "peak": peak, "unit": "G wave-instr/s",
                                     "frac": ach / peak, "valu_wave_insts_per_launch": pm["valu_insts"],
                                     "waves_per_simd": waves, "attainable_at_this_occupancy": attainable,
                                     "frac_of_attainable": ach / attainable,
                                     # ... and in CYCLES (no wall clock in it): SIMD cycles per vector instruction at the clock this run ran at
                                     "cycles_per_simd_inst": ((main_k_ms * 1e-3) * clock["mean"] * 1e6 * (min(n, 256 * 4) if lone else 256 * 4) / pm["valu_insts"]) if clock else None,
                                     "attainable_cycles_per_simd_inst": {1: 4.44, 2: 4.0, 4: 3.1, 5: 2.5, 8: 2.55}[waves],
                                     "attainable_source": "profiles/r03_valu_issue_probe.txt (mix_add_max row at this occupancy)",
                                     "source": pm["source"], "stale": pm["stale"]}
            out["roofline"] = roof
        if not dry and world == 1:
            extras(args, out, w, L, al, blob, q_off, q_len, t_off, t_len, rec, n)
    if al is not None:
        w.RecycleAligner(al)
        al = None
    if rank == 0:
        if not dry and world == 1 and args.other_configs:
            # the other configurations, driver-observed: short legs on the same GPU after the headline's timed region (its
            # context and buffers released first: the configs[4] sample alone takes 60 % of HBM for its arenas)
            del d_blob, d_qoff, d_toff, d_qlen, d_tlen, d_rec, d_ops
            torch.cuda.empty_cache()
            out["config"]["other_configs"] = other_configs(w, L, torch, dev, dev_index)
        print(json.dumps(out), flush=True)
    if coll:
        dist.barrier()
        dist.destroy_process_group()


KNAMES = ["wfa_generic_kernel", "wfa_packed_kernel", "wfa_reg_kernel<2, 4, 1>", "wfa_blk_kernel<16",
          "wfa_blk_kernel<8", "wfa_blk_kernel<64", "wfa_blk_kernel<8, 8, false, 4", "wfa_team_kernel", "wfa_duo_kernel", "wfa_blk_kernel<32", "wfa_lane_kernel",
          "wfa_blk_kernel<16, 1, false, 0, false, true, false>", "wfa_blk_kernel<32, 1, false, 0, true, true, false>",
          "wfa_blk_kernel<64, 1, false, 0, false, true, false>", "wfa_blk_kernel<64, 1, false, 1, false, true, false>",
          "wfa_blk_kernel<64, 1, false, 2, false, true, false>",
          "wfa_blk_kernel<64, 1, false, 1, false, false",  # 16: the lone-pair instance of wfahip_align_pair
          "wfa_teamc_kernel",                              # 17: wide wavefronts, one backtrace word per diagonal (round 5)
          "wfa_wide_kernel"]                               # 18: semi-global reads up to 2 047 bases, a workgroup per pair with the rows in LDS rings (round 6)
# legs of config.other_configs: (config, timed steps, warm-up steps)
# (round 5: every configuration that had a line only under profiles/ now has a driver-observed leg -- about two minutes in all)
OTHER_LEGS = [("c2", 300, 5), ("c2m", 100, 3), ("p242", 12, 3), ("g3", 3, 1), ("k10", 30, 4), ("k20", 12, 3), ("l5", 12, 3), ("l20", 5, 2), ("L5", 4, 2),
              ("c4", 3, 1), ("c5s", 2, 1), ("c5s32", 1, 1)]


def other_configs(w, L, torch, dev, dev_index):
    """Short legs of the other configurations (BASELINE configs[1], the reference's 10 % and 50 kbp rows, the configs[4]
    sample): the same step as the headline -- sequences resident in HBM -> records + CIGAR ops in HBM through
    wfahip_align_batch_device --, a few steps each, with the dominant kernel's share of the HBM roofline computed the same way
    (algorithmic bytes from an untimed census pass / its launches' duration in a step).  One failing leg does not take
    the headline line with it: it is reported as {"error": ...}."""
    import ctypes as C
    import numpy as np
    res = {}
    for name, steps, warmup in OTHER_LEGS:
        c = CONFIGS[name]
        t_leg = time.perf_counter()
        al = d = d_rec = d_ops = step = None
        try:
            n = c["pairs"] or c["total"]  # (c4: all 1e7 pairs on the one GPU)
            pen = tuple(c.get("pen", (4, 6, 2)))
            al = w.New(w.Penalties(*pen), w.Options(GlobalAlignment=not c["semi_global"]), device=dev_index)
            if c["adaptive"]:
                assert al.AdaptiveReduction(w.DefaultAdaptiveOption) is None
            if n >= 5_000_000:  # generated in HBM, byte for byte the host generator's dataset (wfahip_generate_pairs_device): 20 GB never cross PCIe
                d = list(w.generate_pairs_device(al, c["seed"], n, c["length"], c["error"]))
                q_len, t_len = d[2].cpu().numpy().view(np.uint32), d[4].cpu().numpy().view(np.uint32)
                blob_size = int(d[0].numel())
            else:
                blob, q_off, q_len, t_off, t_len = w.generate_pairs(c["seed"], n, c["length"], c["error"], n_threads=min(32, os.cpu_count() or 8))
                d = [torch.from_numpy(a).to(dev) for a in (blob, q_off.view(np.int64), q_len.view(np.int32), t_off.view(np.int64), t_len.view(np.int32))]
                blob_size = blob.size
            max_len = int(max(q_len.max(), t_len.max()))
            sum_len = int(q_len.astype(np.int64).sum() + t_len.astype(np.int64).sum())
            ops_cap = int(sum_len * max(0.25, 3.0 * c["error"])) + 8 * n + 1024
            if c["semi_global"] or c["length"] >= 20000:
                ops_cap = sum_len + 2 * n + 1024
            d_rec = torch.zeros((n, L.REC_WORDS), dtype=torch.int32, device=dev)
            d_ops = torch.zeros(ops_cap, dtype=torch.int64, device=dev)
            prm, lib, timing = al._params(), L.lib(), L.Timing()
            stream = torch.cuda.current_stream(dev).cuda_stream

            def step():
                needed = C.c_uint64()
                L.check(lib.wfahip_align_batch_device(al._ctx, C.byref(prm), d[0].data_ptr(), blob_size, d[1].data_ptr(), d[2].data_ptr(),
                                                      d[3].data_ptr(), d[4].data_ptr(), n, max_len, d_rec.data_ptr(), d_ops.data_ptr(),
                                                      ops_cap, C.byref(needed), stream), f"wfahip_align_batch_device ({name})")
                lib.wfahip_last_timing(al._ctx, C.byref(timing))

            for _ in range(warmup):
                step()
            torch.cuda.synchronize(dev)
            main_ms = []
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
                main_ms.append(timing.main_kernel_ms)
            torch.cuda.synchronize(dev)
            elapsed = time.perf_counter() - t0
            kind = int(timing.main_kernel_kind)
            L.check(lib.wfahip_set_option(al._ctx, b"census", 1), "census")
            step()
            torch.cuda.synchronize(dev)
            rec = d_rec.cpu().numpy().view(np.uint32)
            cells = int(rec[:, L.REC_CELLS_LO].astype(np.uint64).sum()) + (int(rec[:, L.REC_CELLS_HI].astype(np.uint64).sum()) << 32)
            n_ops = int(rec[:, L.REC_OPS_LEN].astype(np.uint64).sum())
            alg_bytes = 4 * cells + sum_len + 64 * n + 8 * n_ops
            mk = float(np.mean(main_ms))
            res[name] = {"workload": f"{n} x {c['length']} bp @{c['error']:.0%}, {'semi-global' if c['semi_global'] else 'global'} {'/'.join(str(v) for v in pen)}, "
                                     f"wf-adaptive {'10/50/1' if c['adaptive'] else 'off'}, seed {c['seed']}",
                         "value": n * steps / elapsed, "unit": "pairs/s", "steps": steps, "warmup": warmup,
                         "ms_per_step": elapsed / steps * 1e3, "kernel": KNAMES[min(kind, len(KNAMES) - 1)], "kernel_ms": mk,
                         "roofline": {"bound": "hbm", "frac": (alg_bytes / (mk * 1e-3) / 8e12) if mk > 0 else None,
                                      "algorithmic_bytes_per_launch": alg_bytes, "peak": 8000.0, "unit": "GB/s"},
                         "status_ok": int((rec[:, L.REC_STATUS] == 0).sum()), "pairs": n,
                         "retried_pairs": int(timing.n_retried_pairs), "leg_s": None}
        except Exception as e:  # noqa: BLE001 -- reported, not raised: the headline line must still be printed
            res[name] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            if al is not None:
                w.RecycleAligner(al)
            d = d_rec = d_ops = step = None  # (a failed leg must not keep its tensors alive into the next one)
            torch.cuda.empty_cache()
        res[name]["leg_s"] = time.perf_counter() - t_leg
    return res


def gpu_telemetry(dev_index):
    """The GPU's clocks, power and temperature as amdgpu's sysfs files give them, or the error that kept them from being
    read.  No child process: under `rocprofv3 --pmc` a child (`rocm-smi` is a `#!/usr/bin/env python3` script) would inherit the
    profiler's preload, initialise the GPU inside `env` and then exec -- the hop this pool forbids."""
    import glob
    try:
        cards = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"), key=lambda d: int(re.search(r"card(\d+)", d).group(1))):
            try:
                if open(os.path.join(dev, "vendor")).read().strip() == "0x1002" and os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
                    cards.append(dev)
            except OSError:
                continue
        if not cards:
            return {"error": "no amdgpu device under /sys/class/drm"}
        dev = cards[min(dev_index, len(cards) - 1)]
        keep = {"sysfs": dev}

        def current_level(name):
            for line in open(os.path.join(dev, name)).read().splitlines():
                if line.rstrip().endswith("*"):
                    return line.split(":", 1)[1].replace("*", "").strip()
            return None
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
            try:
                keep[name] = current_level(name)
            except OSError:
                pass
        try:
            keep["performance_level"] = open(os.path.join(dev, "power_dpm_force_performance_level")).read().strip()
        except OSError:
            pass
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            for f, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6), ("temp1_input", "temperature_edge_c", 1e-3),
                                  ("temp2_input", "temperature_junction_c", 1e-3), ("temp3_input", "temperature_mem_c", 1e-3), ("freq1_input", "sclk_hz", 1.0)):
                try:
                    keep.setdefault(key, round(int(open(os.path.join(hw, f)).read().strip()) * scale, 1))
                except (OSError, ValueError):
                    pass
        return keep
    except Exception as e:  # noqa: BLE001 -- telemetry must never take the bench line with it
        return {"error": f"{type(e).__name__}: {e}"}


def find_profile(config, kname):
    """Newest profiles/*_<config>_pmc.json (or, for c3, *_pmc_hbm.json) that holds the dominant kernel."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{config}_pmc.json")), reverse=True)
    if config == "c3":
        cands += sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")), reverse=True)
    src_sha = kernel_sources_sha()
    for f in cands:
        try:
            doc = json.load(open(f))
            pm = doc["kernels"]
        except Exception:
            continue
        # (the blocked kernel has two instances: the timed steps run the one without the census -- its fifth template argument)
        import re
        no_census = lambda nm: 0 if re.search(r"wfa_blk_kernel<\d+, \d+, \w+, \d+, false", nm) else 1
        # (names of before the seventh template argument, LDSA, are prefixes of today's up to the closing bracket)
        kpre = kname[:-1] if kname.endswith(">") else kname
        for name, grids in sorted(pm.items(), key=lambda kv: no_census(kv[0])):
            if kname in name or (kpre + ", ") in name:
                g0 = max(grids.values(), key=lambda d: d.get("WRITE_SIZE_KB", 0))
                if "FETCH_SIZE_KB" in g0 and "WRITE_SIZE_KB" in g0:
                    return {"traffic": (2.0 * g0["FETCH_SIZE_KB"] + g0["WRITE_SIZE_KB"]) * 1024.0,
                            "valu_insts": g0.get("SQ_INSTS_VALU"), "source": os.path.basename(f),
                            "kernel": name.replace("void wfa::", "").split("(")[0],
                            "stale": doc.get("kernel_sources_sha") != src_sha}
    return None


def kernel_sources_sha():
    """Hash of wfa_amd/csrc + include: tells whether a committed PMC profile belongs to this build of the kernels."""
    import hashlib
    h = hashlib.sha1()
    for d in ("wfa_amd/csrc", "include"):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            with open(os.path.join(ROOT, d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def extras(args, out, w, L, al, blob, q_off, q_len, t_off, t_len, rec, n):
    """N = 1 only, after the timed region: the CPU baseline (oracle = C restatement of the reference) on one core and
    on all host cores, the host-to-host entry (wfahip_align_batch: what the cgo binding calls), single-pair latency."""
    import numpy as np
    cfg = out["config"]
    if args.cpu_sample > 0:
        from oracle import oracle as O
        ns = min(args.cpu_sample, n)
        p = O.make_params(*args.pen, global_alignment=not args.semi_global, adaptive=None if args.no_adaptive else (10, 50, 1))
        t1 = time.perf_counter()
        ref = O.align_batch(p, blob, q_off[:ns], q_len[:ns], t_off[:ns], t_len[:ns], n_threads=args.cpu_threads,
                            want_ops=False)
        dt = time.perf_counter() - t1
        same = bool(np.array_equal(ref.score, rec[:ns, L.REC_SCORE]) and
                    np.array_equal(ref.align_len, rec[:ns, L.REC_ALIGN_LEN]))
        out["cpu_baseline"] = {"value": ns / dt, "unit": "pairs/s", "cores": args.cpu_threads, "kind": "port",
                               "sample": f"first {ns} pairs of the same dataset, oracle/wfa_oracle.c "
                                         f"(C restatement of the Go reference; Go itself is not installed), "
                                         f"{dt:.1f} s", "scores_match_gpu": same,
                               "published_reference": "6483 pairs/s (wfa-go, laptop, 1 thread; README.md:330)"}
        if args.cpu_all_cores >= 2 or (args.cpu_all_cores == 1 and args.length < 20000):  # (long reads: minutes per pair)
            cores = os.cpu_count() or 1
            na = min(n, max(ns, int(ns / dt * 0.5 * cores * 8)))  # ~8 s of work if it scaled at 50 %
            t1 = time.perf_counter()
            ref = O.align_batch(p, blob, q_off[:na], q_len[:na], t_off[:na], t_len[:na], n_threads=cores, want_ops=False)
            dta = time.perf_counter() - t1
            out["cpu_baseline"]["all_cores"] = {"value": na / dta, "unit": "pairs/s", "cores": cores,
                                                "sample": f"first {na} pairs, one oracle aligner per thread, {dta:.1f} s",
                                                "scores_match_gpu": bool(np.array_equal(ref.score, rec[:na, L.REC_SCORE]))}
    if args.host_entry:
        # SURVEY.md section 8d timing (b): host byte blobs -> host result arrays through wfahip_align_batch
        # (PCIe-inclusive; never `value`).  First call allocates staging buffers: untimed.
        import ctypes as C
        prm = al._params()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        ts = []
        # (two untimed calls first: the first allocates staging buffers and the page-locked packing buffer, and the result
        # blocks only become page-locked when they come back from wfahip_results_free the first time)
        reps = 3
        for i in range(reps + 2):
            res = L.Results()
            t1 = time.perf_counter()
            L.check(L.lib().wfahip_align_batch(al._ctx, C.byref(prm), vp(blob), blob.size, vp(q_off), vp(q_len), vp(t_off),
                                               vp(t_len), n, C.byref(res)), "wfahip_align_batch")
            dt = time.perf_counter() - t1
            if i == reps + 1:
                sc = np.ctypeslib.as_array(res.score, shape=(n,))
                cfg["host_to_host_scores_match_device_entry"] = bool(np.array_equal(sc, rec[:, L.REC_SCORE]))
            L.lib().wfahip_results_free(C.byref(res))
            if i > 1:
                ts.append(dt * 1e3)
        # ... and from pre-packed 2-bit input (wfahip_align_batch_packed: a quarter of the bytes cross PCIe); the packing
        # itself (wfahip_pack_pairs, host threads) is timed separately
        if args.length <= 20000:
            t1 = time.perf_counter()
            packed, q_woff, t_woff = w.pack_pairs(blob, q_off, q_len, t_off, t_len, n_threads=min(32, os.cpu_count() or 8))
            cfg["host_pack_ms"] = (time.perf_counter() - t1) * 1e3
            tp = []
            for i in range(reps + 1):
                res = L.Results()
                t1 = time.perf_counter()
                L.check(L.lib().wfahip_align_batch_packed(al._ctx, C.byref(prm), vp(packed), packed.size, vp(q_woff), vp(q_len),
                                                          vp(t_woff), vp(t_len), n, C.byref(res)), "wfahip_align_batch_packed")
                dt = time.perf_counter() - t1
                if i == reps:
                    sc = np.ctypeslib.as_array(res.score, shape=(n,))
                    cfg["host_to_host_packed_scores_match"] = bool(np.array_equal(sc, rec[:, L.REC_SCORE]))
                L.lib().wfahip_results_free(C.byref(res))
                if i > 0:
                    tp.append(dt * 1e3)
            cfg["host_to_host_packed_ms"] = min(tp)
        cfg["host_to_host_ms"] = min(ts)
        cfg["host_to_host_pairs_per_s"] = n / (min(ts) * 1e-3)
        cfg["host_to_host_note"] = ("wfahip_align_batch: pageable host blobs -> malloc'd host result arrays, "
                                    f"min of {reps} calls after two untimed calls; large pure-ACGT batches are 2-bit packed by host threads "
                                    "slice by slice beside the upload (included); PCIe-inclusive, never `value`")
    if args.latency:
        # single-pair round trip through the drop-in API (Aligner.Align = a batch of one)
        k = min(n, 200)
        qs = [bytes(blob[int(q_off[i]):int(q_off[i]) + int(q_len[i])]) for i in range(k)]
        tsq = [bytes(blob[int(t_off[i]):int(t_off[i]) + int(t_len[i])]) for i in range(k)]
        al.Align(qs[0], tsq[0])
        t1 = time.perf_counter()
        for i in range(k):
            al.Align(qs[i], tsq[i])
        cfg["single_pair_align_us"] = (time.perf_counter() - t1) / k * 1e6
        cfg["single_pair_note"] = f"mean over {k} Aligner.Align calls (python ctypes mirror; batch of one per call)"


if __name__ == "__main__":
    main()
