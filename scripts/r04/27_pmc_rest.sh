#!/bin/bash
# Round 4: the PMC passes of scripts/r04/20_profiles.sh for every workload but config 3 (whose passes scripts/r04/26_c3_pmc.sh redid with
# k_din_fused's two forms kept apart -- and whose summary, written on the GPU box, replaced the first run's file when it was merged
# back).  Writes gpurun_out/r04_prof/pmc_summary_rest.json; merge with pmc_summary_c3.json locally.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_prof
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads="
declare -A WL
WL[c2]="--input-batches 32"
WL[c2_hbm]="--big-vocab 8388608 --input-batches 32"
WL[c2_pairs]="--workload deepfm_c2"
WL[c4_pairs]="--workload deepfm_c4"
WL[c5]="--workload widedeep_c5"
WL[v2_ref]="--workload deepfm_v2_ref"
WL[din_ref]="--workload din_ref"
cd /tmp && export TMPDIR=/tmp
pass() { tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- "$@" > $O/pmc_$tag.log 2>&1; }
for w in c2 c2_hbm c2_pairs c4_pairs c5 v2_ref din_ref; do
  CMD="python $R/bench.py ${WL[$w]} --steps 20 --warmup 5 $STRICT"
  pass ${w}_fetch FETCH_SIZE -- $CMD
  pass ${w}_write WRITE_SIZE -- $CMD
done
for w in c2 c2_pairs c5; do
  CMD="python $R/bench.py ${WL[$w]} --steps 20 --warmup 5 $STRICT"
  pass ${w}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $CMD
  pass ${w}_sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -- $CMD
done
cd $R
python - <<'PY'
import csv, glob, collections, os, json
summary = {}
for d in sorted(glob.glob('gpurun_out/r04_prof/pmc_*/')):
    tag = os.path.basename(d.rstrip('/'))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if '(anonymous namespace)::' not in k: continue
            short = k.split('(anonymous namespace)::')[1].split('(')[0]
            short = short if short.startswith('k_din_fused<') else short.split('<')[0]
            if any(s in short for s in ('prep', 'fold', 'absmax', 'split', 'pack', 'build', 'count_small', 'swizzle', 'coef')): continue
            summary.setdefault(tag, {})[short] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
            summary[tag][short]['launches'] = len(next(iter(cs.values())))
json.dump(summary, open('gpurun_out/r04_prof/pmc_summary_rest.json', 'w'), indent=1, sort_keys=True)
print(len(summary), 'pmc passes summarised')
PY
rm -rf $O/pmc_*/
