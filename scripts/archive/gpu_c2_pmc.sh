#!/bin/bash
# config-2 parity + bench, then the FETCH_SIZE / WRITE_SIZE PMC passes of the same command (separate runs).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash scripts/gpu_c2.sh
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c2_$c -o p -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-check > /dev/null 2>&1
python - <<PY
import csv,glob
v=[float(r['Counter_Value']) for f in glob.glob('$R/gpurun_out/pmc_c2_$c/**/*counter_collection.csv',recursive=True) for r in csv.DictReader(open(f)) if 'k_deepfm_v2_joint' in r['Kernel_Name']]
print("$c KiB per launch avg", sum(v)/len(v), "n", len(v))
PY
done
