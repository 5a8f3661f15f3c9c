#!/bin/bash
# VGPR / SGPR / scratch / occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage; device pass only).
# usage: scripts/kernel_resources.sh [filter-regex]  -> build/kernel_resources.txt
cd "$(dirname "$0")/.."
mkdir -p build
[ -n "$SKIP_COMPILE" ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I sparrowrecsys_amd/csrc -DSPRK_SINGLE_TU --cuda-device-only -c \
  -Rpass-analysis=kernel-resource-usage sparrowrecsys_amd/csrc/sparrow_hip.hip -o /dev/null 2> build/kernel_resources.raw
python3 - build/kernel_resources.raw "${1:-.}" <<'PY' | tee build/kernel_resources.txt
import re, sys, subprocess
pat = re.compile(sys.argv[2])
cur = None
rows = {}
for line in open(sys.argv[1]):
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[.*?\])?(?: \[.*\])?):\s*(\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
for n, d in zip(names, dem):
    d = re.sub(r"^void ", "", d).replace("(anonymous namespace)::", "").replace("sprk_dev::", ""); d = re.sub(r"\(.*$", "", d)
    if not pat.search(d): continue
    r = rows[n]
    print("%-84s vgpr %4s agpr %3s sgpr %4s scratch %5s occ %2s lds %6s" % (d[:84], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
          r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
PY
