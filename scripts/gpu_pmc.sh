#!/bin/bash
# PMC counter passes (separate from kernel-trace stats runs), per MI355X_MICROARCH.md guidance.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-deepfm_v2_c2}
EXTRA=${2:-}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCC_MISS_sum|TCC_EA0_RDREQ_sum|TCC_EA0_RDREQ_32B_sum|TCC_REQ_sum|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_VALU_MFMA_BUSY_CYCLES|SQ_INSTS_VALU_MFMA_MOPS_F32|SQ_INSTS_MFMA|SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_LDS|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_INST_CYCLES_VMEM|SQ_WAIT_INST_LDS|GRBM_GUI_ACTIVE|TCP_TCC_READ_REQ_sum|TCP_TOTAL_CACHE_ACCESSES_sum|TCC_BUSY_sum)\b" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_available.txt
cat $R/gpurun_out/pmc_available.txt; echo
run() { # name counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${WL}_$name -o p -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-check --workload $WL $EXTRA > $R/gpurun_out/pmc_${WL}_$name.log 2>&1
}
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run write WRITE_SIZE TCC_REQ_sum
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
cd $R
python - <<'PY'
import csv, glob, collections, os, sys
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'][:60]
            agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if 'rocclr' in k: continue
            print(os.path.basename(d.rstrip('/')), k[:50], {c: round(sum(v)/len(v), 1) for c, v in cs.items()}, 'n=%d' % len(next(iter(cs.values()))))
PY
