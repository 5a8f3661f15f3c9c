#!/bin/bash
# SQ-side PMC passes for one bench.py configuration: gpu_pmc2.sh <tag> <bench args...>   (env passes through)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_${TAG}_$name -o p -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-check $BARGS > $R/gpurun_out/pmc2_${TAG}_$name.log 2>&1
}
BARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_WAVES
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT
cd $R
python - <<PY
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc2_${TAG}_*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name'][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if 'chain' not in k: continue
            print(os.path.basename(d.rstrip('/')), {c: round(sum(v)/len(v), 1) for c, v in cs.items()}, 'n=%d' % len(next(iter(cs.values()))))
PY
