#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
C3="python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-check --workload din_c3"
pass() { name=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c3_$name -o p -- "$@" > /dev/null 2>&1; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $C3
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -- $C3
pass sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM -- $C3
pass fetch FETCH_SIZE -- $C3
pass write WRITE_SIZE -- $C3
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc_c3_*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if 'k_din_attn' in k or 'k_din_tail<' in k:
                print(os.path.basename(d.rstrip('/')), k.split('(')[0][-40:], {c: round(sum(v)/len(v), 1) for c, v in cs.items()})
PY
