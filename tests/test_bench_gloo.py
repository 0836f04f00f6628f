"""bench.py's own multi-rank branch, run on CPU: `--gpus 2` with WORLD_SIZE unset makes bench.py start the two ranks
itself (a torch.distributed.run child), `--backend gloo --dry` replaces the kernels by a deterministic record per pair
so that everything else -- process-group init, barriers, the asynchronous record gather after every step, the
max-over-ranks reduction of the elapsed time, the JSON line -- is the code the GPU run executes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, timeout=280):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--dry", "--length", "60",
                        "--steps", "3", "--warmup", "1"] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_gpus_flag_starts_the_ranks_weak(built):
    out = _run(["--gpus", "2", "--pairs", "1000"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    c = out["config"]
    assert c["ranks_seen_by_collective"] == 2 and c["pairs_per_rank"] == [1000, 1000]
    assert c["total_pairs_per_step"] == 2000 and c["gathered_records_complete"] is True
    assert c["gather_bytes_per_rank_step"] == 1000 * 11 * 4 and c["gather_ms_standalone"] > 0
    assert out["steps"] == 3 and out["value"] == pytest.approx(2000 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-6)


@pytest.mark.timeout(300)
def test_total_pairs_is_strong_scaling_with_uneven_shards(built):
    out = _run(["--gpus", "2", "--total-pairs", "2001", "--gather-ops"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    c = out["config"]
    assert c["pairs_per_rank"] == [1001, 1000] and c["total_pairs_per_step"] == 2001
    assert c["gathered_records_complete"] is True and c["gather"] == "records + CIGAR ops"


@pytest.mark.timeout(300)
def test_single_rank_dry_line(built):
    out = _run(["--gpus", "1", "--pairs", "500"])
    assert out["n_gpus"] == 1 and out["config"]["pairs_per_rank"] == [500] and out["config"]["dry"] is True


@pytest.mark.timeout(300)
def test_force_collective_with_one_rank(built):
    """`--gpus 1 --force-collective`: the process group exists with ONE rank and every collective of the multi-rank path runs
    (here over gloo; on the GPU box the same flag runs them over RCCL: tests/test_scale_gpu.py)."""
    out = _run(["--gpus", "1", "--force-collective", "--pairs", "700"])
    c = out["config"]
    assert out["n_gpus"] == 1 and c["backend"] == "gloo" and c["collective_forced"] is True
    assert c["pairs_per_rank"] == [700] and c["gathered_records_complete"] is True and c["gather_ms_standalone"] > 0
    assert c["gather_bytes_per_rank_step"] == 700 * 11 * 4


@pytest.mark.timeout(600)
def test_eight_ranks_uneven_total(built):
    """Pre-flight of the 8-GPU run the driver makes (configs[3]: pairs sharded over the ranks, shards that differ by one pair):
    eight ranks over gloo, --total-pairs not divisible by eight, the records of every shard gathered in pair order."""
    out = _run(["--gpus", "8", "--total-pairs", "100003", "--length", "20"], timeout=560)
    c = out["config"]
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and c["ranks_seen_by_collective"] == 8
    assert c["pairs_per_rank"] == [12501, 12501, 12501, 12500, 12500, 12500, 12500, 12500] and c["total_pairs_per_step"] == 100003
    assert c["gathered_records_complete"] is True


def test_plan_prints_the_per_rank_footprint():
    """`bench.py --plan`: no GPU, no child processes -- what each rank of the configs[3] run holds in HBM, and whether it fits."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plan", "--gpus", "8", "--config", "c4"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr[-1000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 8 and d["pairs_total"] == 10_000_000 and d["all_fit"] is True and len(d["ranks"]) == 8
    assert [r["pairs"] for r in d["ranks"]] == [1_250_000] * 8 and all(r["total_gib"] < 288 * 0.95 for r in d["ranks"])
    # all ten million pairs on ONE GPU: the chunks alternate through one arena buffer, and it still fits
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plan", "--gpus", "1", "--config", "c4"], capture_output=True, text=True, timeout=60)
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0 and d["ranks"][0]["chunks"] >= 3 and d["all_fit"] is True
    # a device too small for a shard says so (exit code 3)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plan", "--gpus", "1", "--config", "c4", "--hbm-gib", "24"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and json.loads(p.stdout.strip().splitlines()[-1])["all_fit"] is False


def test_configs_name_the_baseline_workloads():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args([])
    assert (a.pairs, a.length, a.error, a.seed, a.semi_global, a.no_adaptive, a.total_pairs) == (1_000_000, 1000, 0.05, 3, False, False, 0)
    a = bench.parse_args(["--config", "c2"])
    assert (a.pairs, a.length, a.error, a.seed, a.no_adaptive) == (100_000, 150, 0.02, 2, True)
    a = bench.parse_args(["--config", "c4", "--gpus", "8"])
    assert (a.total_pairs, a.length, a.seed) == (10_000_000, 1000, 4)
    a = bench.parse_args(["--config", "c5s"])
    assert (a.pairs, a.length, a.error, a.semi_global, a.no_adaptive) == (8, 100_000, 0.10, True, False)
    # the default run also measures short legs of the other configurations (config.other_configs); any other run does not
    assert bench.parse_args([]).other_configs == 1 and bench.parse_args(["--config", "c2"]).other_configs == 0
    assert bench.parse_args(["--pairs", "1000"]).other_configs == 0 and bench.parse_args(["--gpus", "2"]).other_configs == 0
    # (round 5: every configuration that used to have a line only under profiles/ is a driver-observed leg)
    legs = [name for name, _, _ in bench.OTHER_LEGS]
    assert len(legs) >= 10 and {"c2", "c2m", "p242", "k10", "k20", "l5", "l20", "L5", "c4", "c5s", "c5s32"} <= set(legs)
    for name in ("c2m", "L5", "k10", "k20", "l5", "l10", "l20", "p242", "g3", "c5s32"):
        assert name in bench.CONFIGS and name in bench.DEFAULT_STEPS
    a = bench.parse_args(["--config", "p242"])
    assert (a.pen, a.pairs, a.length, a.seed, a.semi_global) == ((2, 4, 2), 1_000_000, 1000, 3, False)
    a = bench.parse_args(["--config", "g3"])
    assert (a.pen, a.pairs, a.length, a.seed, a.semi_global) == ((4, 6, 2), 1_000_000, 1000, 3, True)
    assert bench.parse_args(["--penalties", "1,1,1"]).pen == (1, 1, 1)


def test_committed_profiles_hold_the_kernels_the_lines_name():
    """`roofline.traffic` comes from the committed PMC profile of a configuration: the name bench.py looks for (the library's
    main_kernel_kind -> KNAMES) must be found in it, the instance WITHOUT the census of stored words first -- also for names
    that were recorded before the kernel template got its latest argument."""
    sys.path.insert(0, ROOT)
    import bench
    for cfg, kname, census_arg in (("c3", "wfa_duo_kernel<false, 2, 4>", None), ("p242", "wfa_duo_kernel<false, 1, 3>", None),
                                   ("c2", "wfa_lane_kernel", None), ("c2m", "wfa_lane_kernel", None),
                                   ("c5s", "wfa_teamc_kernel", None), ("k10", "wfa_blk_kernel<16", "false"),
                                   ("l5", bench.KNAMES[15], "false"), ("L5", bench.KNAMES[11], "false"), ("l20", bench.KNAMES[13], "false")):
        pm = bench.find_profile(cfg, kname)
        assert pm is not None and pm["traffic"] > 0, (cfg, kname)
        if census_arg is not None:  # wfa_blk_kernel<G, BATCH, STREAM, PPT, CENSUS, ...>: the timed steps run CENSUS = false
            assert pm["kernel"].split(", ")[4].rstrip(">") == census_arg, (cfg, pm["kernel"])
    # a six-argument name of before round 4's LDSA argument still finds today's instance (round 5: nine arguments -- the penalty shape)
    old = bench.KNAMES[15].rsplit(", ", 1)[0] + ">"
    assert bench.find_profile("l5", old)["kernel"].startswith(bench.KNAMES[15][:-1])
    # the instance of the line's own penalty shape, not another shape's that ran in the same profile
    assert "<false, 2, 4>" in bench.find_profile("c3", "wfa_duo_kernel<false, 2, 4>")["kernel"]
