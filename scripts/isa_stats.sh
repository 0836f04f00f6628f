#!/bin/bash
# Static instruction mix of one kernel in build/*.s (after `make asm`).  Usage: scripts/isa_stats.sh <mangled-name-substring>
S=build/asm/wfa_host-hip-amdgcn-amd-amdhsa-gfx950.s
K=${1:-wfa_blk_kernelILi16E}
python3 - "$S" "$K" <<'PY'
import sys, re
src, key = sys.argv[1], sys.argv[2]
out, on = [], False
for line in open(src):
    if not on and line.startswith("_ZN3wfa") and key in line and line.split()[0].endswith(":"):
        on = True
    if on:
        out.append(line)
        if ".end_amdhsa_kernel" in line:
            break
open("build/asm/kernel.s", "w").writelines(out)
body = [l.strip() for l in out]
cnt = lambda pat: sum(1 for l in body if re.match(pat, l))
print("lines", len(body), "VALU", cnt(r"v_"), "SALU", cnt(r"s_"), "cndmask", cnt(r"v_cndmask"), "cmp", cnt(r"v_cmp"),
      "nop", cnt(r"s_nop"), "dpp", sum("_dpp" in l for l in body), "ds", cnt(r"ds_"), "branch", cnt(r"s_cbranch"),
      "waitcnt", cnt(r"s_waitcnt"), "saveexec", sum("saveexec" in l for l in body))
PY
