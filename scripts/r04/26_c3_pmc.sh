#!/bin/bash
# Round 4: config 3's PMC passes again with k_din_fused's instantiations kept apart (the first summary of scripts/r04/20_profiles.sh
# averaged the one-launch kernel and the attention-only loop's kernel into one row); merged into gpurun_out/r04_prof/pmc_summary.json.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_prof
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads="
cd /tmp && export TMPDIR=/tmp
pass() { tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- "$@" > $O/pmc_$tag.log 2>&1; }
CMD="python $R/bench.py --workload din_c3 --steps 20 --warmup 5 $STRICT"
pass c3_fetch FETCH_SIZE -- $CMD
pass c3_write WRITE_SIZE -- $CMD
pass c3_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $CMD
pass c3_sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -- $CMD
cd $R
python - <<'PY'
import csv, glob, collections, os, json
path = 'gpurun_out/r04_prof/pmc_summary.json'
summary = json.load(open(path)) if os.path.exists(path) else {}
for d in sorted(glob.glob('gpurun_out/r04_prof/pmc_c3_*/')):
    tag = os.path.basename(d.rstrip('/'))
    summary[tag] = {}
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if '(anonymous namespace)::' not in k: continue
            short = k.split('(anonymous namespace)::')[1].split('(')[0]
            short = short if short.startswith('k_din_fused<') else short.split('<')[0]
            if any(s in short for s in ('prep', 'fold', 'absmax', 'split', 'pack', 'build', 'count_small', 'swizzle', 'coef')): continue
            summary[tag][short] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
            summary[tag][short]['launches'] = len(next(iter(cs.values())))
json.dump(summary, open(path, 'w'), indent=1, sort_keys=True)
for t in sorted(summary):
    if 'c3' in t: print(t, json.dumps(summary[t])[:700])
PY
rm -rf $O/pmc_c3_*/
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k 'beyond_one_round' 2>&1 | tail -2
