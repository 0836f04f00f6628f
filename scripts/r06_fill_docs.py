"""Round 6, closing: put the measured numbers of the closing run (profiles/r06_bench_lines/, profiles/r06_c3_pmc.json, profiles/r06_gpu_tests.txt) into the
placeholders -- (C3_MS) and the like -- that DESIGN.md, BASELINE.md and README.md carry until then.  Usage: python scripts/r06_fill_docs.py"""
import json, os, re, subprocess, sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = os.path.join(R, "profiles", "r06_bench_lines")


def line(name):
    return json.loads([ln for ln in open(os.path.join(L, f"bench_{name}_n1.json")) if ln.startswith("{")][-1])


c3, g3, c5s, L5, c2 = line("c3"), line("g3"), line("c5s"), line("L5"), line("c2")
r, sec, cb = c3["roofline"], c3["roofline"].get("secondary") or {}, c3["cpu_baseline"]
pmc = json.load(open(os.path.join(R, "profiles", "r06_c3_pmc.json")))
duo = [v for k, v in pmc["kernels"].items() if "wfa_duo_kernel<false" in k][0]
duo = max(duo.values(), key=lambda v: v.get("SQ_INSTS_VALU", 0))
summ = open(os.path.join(R, "profiles", "r06_c3_summary.txt")).read()
m = re.search(r"wfa_duo_kernel<false[^\n]*'AverageNs': '([0-9.]+)'", summ)
tests = open(os.path.join(R, "profiles", "r06_gpu_tests.txt")).read()
tm = re.search(r"(\d+) passed[^\n]* in ([0-9.]+)s", tests)


def e(v):
    s = f"{v:.3g}"
    return s.replace("e+0", "e").replace("e+", "e")


T = {
    "C3_VALUE": e(c3["value"]), "C3_MS": f"{c3['ms_per_step']:.2f}", "C3_FWD": f"{r['kernel_ms']:.2f}", "C3_FRAC": f"{r['frac']:.3f}", "C3_ACH": f"{r['achieved']:.0f}",
    "C3_PROF_MS": f"{float(m.group(1)) / 1e6:.2f}" if m else "?", "C3_TRAFFIC": f"{r['traffic'] / 1e9:.2f}" if r.get("traffic") else "?",
    "C3_TRATIO": f"{r['traffic'] / r['algorithmic_bytes_per_launch']:.2f}" if r.get("traffic") else "?",
    "C3_VALU": e(duo["SQ_INSTS_VALU"]), "C3_GIPS": f"{sec.get('achieved', 0):.0f}", "C3_ATT": f"{sec.get('frac_of_attainable', 0):.2f}",
    "C3_BANK": f"{100.0 * duo['SQ_LDS_BANK_CONFLICT'] / duo['SQ_LDS_IDX_ACTIVE']:.1f}",
    "CPU1": e(cb["value"]), "CPUALL": e(cb["all_cores"]["value"]), "G3_VALUE": e(g3["value"]), "G3_MS": f"{g3['ms_per_step']:.0f}",
    "C5S_VALUE": f"{c5s['value']:.1f}", "C3_H2H": f"{c3['config']['host_to_host_ms']:.1f}", "L5_MS": f"{L5['ms_per_step']:.1f}", "C2_FRAC": f"{c2['roofline']['frac']:.3f}",
    "TESTS_S": f"{float(tm.group(2)):.0f}" if tm else "?", "TESTS_N": tm.group(1) if tm else "?",
}
table = subprocess.run([sys.executable, os.path.join(R, "scripts", "r06_table.py")], capture_output=True, text=True, check=True).stdout
ktable = subprocess.run([sys.executable, os.path.join(R, "scripts", "r06_table.py"), "--kernels"], capture_output=True, text=True, check=True).stdout
for f in ("DESIGN.md", "BASELINE.md", "README.md", "KERNELS.md"):
    p = os.path.join(R, f)
    s = open(p).read()
    for k, v in T.items():
        s = s.replace(f"({k})", v)
    s = s.replace("(ROUND6_TABLE)", table.rstrip("\n")).replace("(KERNEL_BOUNDS_TABLE)", ktable.rstrip("\n"))
    left = sorted(set(re.findall(r"\(([A-Z0-9_]{4,})\)", s)) & (set(T) | {"ROUND6_TABLE", "KERNEL_BOUNDS_TABLE"}))
    open(p, "w").write(s)
    print(f, "filled", "" if not left else f"LEFT: {left}")
print(json.dumps(T, indent=0))
