#!/bin/bash
# Dev tool: registers / scratch / LDS / occupancy of every kernel of the library (clang's kernel-resource-usage remarks).
#   tools/kernel_resources.sh [extra hipcc flags] > profiles/rNN_kernel_resources.txt
cd "$(dirname "$0")/../bayesssm_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC --cuda-device-only -c -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" bssm_api.hip 2>&1 | python3 -c '
import re, sys, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur); continue
    for key, pat in (("sgpr", r"SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("spill_v", r"VGPRs Spill: (\d+)"), ("spill_s", r"SGPRs Spill: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None: cur[key] = int(m.group(1))
print("%-6s %-6s %-8s %-8s %-5s %-7s %s" % ("vgpr", "sgpr", "scratch", "spill_v", "occ", "lds", "kernel"))
for r in sorted(rows, key=lambda r: r["name"]):
    n = re.sub(r"\(.*", "", r["name"]).replace("void bssm::", "").replace("void ", "")
    print("%-6s %-6s %-8s %-8s %-5s %-7s %s" % (r.get("vgpr"), r.get("sgpr"), r.get("scratch"), r.get("spill_v"), r.get("occ"), r.get("lds"), n))
'
