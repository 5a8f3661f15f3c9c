#!/bin/bash
# PMC counter passes (separate from the kernel-trace --stats runs, per MI355X_MICROARCH.md): memory-side traffic of
# the dominant kernels + a calibration of FETCH_SIZE / WRITE_SIZE on a known-byte-count gather, then SQ-side counters.
# Writes gpurun_out/pmc_summary.json (copy to profiles/rNN/) .
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
pass() { # tag name counters -- command...
  tag=$1; name=$2; shift 2; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -o p -- "$@" > $R/gpurun_out/pmc_${tag}_$name.log 2>&1
}
C2="python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-check"
C3="python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-check --workload din_c3"
CAL="python $R/scripts/pmc_calib.py"
for t in c2 c3 cal; do
  case $t in c2) CMD=$C2;; c3) CMD=$C3;; cal) CMD=$CAL;; esac
  pass $t fetch FETCH_SIZE -- $CMD
  pass $t write WRITE_SIZE TCC_REQ_sum -- $CMD
  pass $t tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -- $CMD
done
for t in c2 c3; do
  case $t in c2) CMD=$C2;; c3) CMD=$C3;; esac
  pass $t sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $CMD
  pass $t sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -- $CMD
done
cd $R
python - <<'PY'
import csv, glob, collections, os, json
summary = {}
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    tag = os.path.basename(d.rstrip('/'))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if 'rocclr' in k or 'at::' in k or 'prep' in k or 'fold' in k or 'absmax' in k or 'split' in k or 'pack' in k:
                continue
            short = k.split('(anonymous namespace)::')[1].split('(')[0].split('<')[0] if '(anonymous namespace)::' in k else k[:40]
            summary.setdefault(tag, {})[short] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
            summary[tag][short]['launches'] = len(next(iter(cs.values())))
json.dump(summary, open('gpurun_out/pmc_summary.json', 'w'), indent=1, sort_keys=True)
for tag, ks in summary.items():
    for k, v in ks.items():
        print(tag, k, v)
PY
grep -h "calib D" gpurun_out/pmc_cal_fetch.log
