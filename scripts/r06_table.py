"""The measurement table of BASELINE.md section 4 (and, with --kernels, the per-kernel table of DESIGN.md section 4) from the bench lines of the closing
run (profiles/r06_bench_lines/).  Usage: python scripts/r06_table.py [--kernels] [dir]"""
import glob, json, os, sys

KERNELS = "--kernels" in sys.argv
if KERNELS:
    sys.argv.remove("--kernels")
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_bench_lines")
ORDER = ["c3", "p242", "g3", "c4", "c2", "c2m", "c5s", "c5s32", "k10", "k20", "l5", "l10", "l20", "L5", "c3_nccl1"]


def sig(v, n=3):
    return "" if v is None else f"{v:.{n}g}"


BOUND = {  # what bounds the dominant kernel of a configuration (DESIGN.md sections 4-6, KERNELS.md)
    "c3": "vector-instruction issue (0.64e12 wave-instructions/s; DPP / select / compare work of WF_NEXT and the band)",
    "p242": "as c3 (a shallower ring, fewer scores)",
    "g3": "first launch (wide rows, 143 ms): the slowest of a pair's four waves in every row; second launch (narrow tail, 77 ms): vector + scalar issue, a pair per wave",
    "c4": "as c3, five chunks",
    "c2": "latency of a wave and a half per SIMD, each as long as the slowest of its 64 pairs",
    "c2m": "vector issue; lanes wait for the slowest pair of their generation",
    "c5s": "the team's row latency (~6 us: two L2 exchanges + the cells of a stripe) x ~53 000 rows per pair",
    "c5s32": "as c5s, every team busy",
    "k10": "vector issue (16 lanes per pair)", "k20": "vector issue (32 lanes per pair)",
    "l5": "lone-wave step latency (500 pairs: half the SIMDs hold one wave)", "l10": "as l5", "l20": "as l5",
    "L5": "vector issue in the first pass, lone-wave latency in the re-run of the 21 % that leave the window",
}
if KERNELS:
    print("| configuration | dominant kernel | algorithmic bytes per pair (section 5) | its launches per step | share of the 8 TB/s HBM roofline | what bounds it |")
    print("|---|---|---|---|---|---|")
    for name in ORDER:
        f = os.path.join(d, f"bench_{name}_n1.json")
        if not os.path.exists(f) or name == "c3_nccl1":
            continue
        j = json.loads([ln for ln in open(f) if ln.startswith("{")][-1])
        c, r = j["config"], j["roofline"]
        n = c.get("total_pairs_per_step") or c.get("pairs_per_gpu")
        b = r['algorithmic_bytes_per_launch'] / n
        bs = f"{b / 1e3:.1f} KB" if b < 1e6 else f"{b / 1e6:.1f} MB" if b < 1e9 else f"{b / 1e9:.1f} GB"
        print(f"| {name} | `{r['kernel']}` | {bs} | {r['kernel_ms']:.4g} ms | {r['frac']:.3f} | {BOUND.get(name, '')} |")
    sys.exit(0)
rows = []
for name in ORDER:
    f = os.path.join(d, f"bench_{name}_n1.json")
    if not os.path.exists(f):
        continue
    lines = [ln for ln in open(f) if ln.startswith("{")]
    if not lines:
        continue
    j = json.loads(lines[-1])
    c, r, cb = j["config"], j["roofline"], j.get("cpu_baseline") or {}
    extra = []
    if c.get("host_to_host_ms"):
        extra.append(f"host bytes -> host results {c['host_to_host_ms']:.1f} ms" + (f", caller-packed {c['host_to_host_packed_ms']:.1f}" if c.get("host_to_host_packed_ms") else ""))
    if c.get("single_pair_align_us"):
        extra.append(f"one `Align` {c['single_pair_align_us']:.0f} us")
    if c.get("retried_pairs"):
        extra.append(f"{c['retried_pairs']} pairs re-run")
    rows.append(f"| {name} | {c['workload']} | {j['value']:.3g} | {j['ms_per_step']:.4g} | `{r['kernel']}` {r['kernel_ms']:.4g} ms | {r['frac']:.3f}"
                + (f" (traffic {r['traffic'] / 1e9:.1f} GB of {r['algorithmic_bytes_per_launch'] / 1e9:.1f} algorithmic)" if r.get("traffic") else "")
                + (("; " + "; ".join(extra)) if extra else "")
                + f" | {sig(cb.get('value'))} | {sig((cb.get('all_cores') or {}).get('value'))} |")
print("| cfg | workload | pairs/s (device-resident, `value`) | ms / step | dominant kernel | HBM fraction (alg. bytes / kernel / 8 TB/s) | CPU port, 1 core | CPU port, all threads |")
print("|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
