#!/bin/bash
# run T: PMC pass over the device tokenizer
set -u
R=$(pwd)
O=$R/gpurun_out/r02t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc1 -o p -- python $R/scripts/bench_ingest.py --rows 20000000 --threads 128 --device > $O/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -o p -- python $R/scripts/bench_ingest.py --rows 20000000 --threads 128 --device > $O/pmc2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ('pmc1','pmc2'):
    for f in glob.glob('gpurun_out/r02t/%s/**/*counter_collection.csv' % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if 'csv' in k:
                print(k[:60], {c: round(sum(v)/len(v)) for c, v in cs.items()})
PY
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +1M -delete
