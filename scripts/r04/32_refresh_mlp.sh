#!/bin/bash
# Round 4: config 5's and EmbeddingMLP.py's evidence again on the final tree (k_mlp_rows changed after scripts/r04/20_profiles.sh ran):
# strict traces, untraced twins, config 5's PMC passes, the GPU suite and the driver's command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_prof
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads="
declare -A WL
WL[c5]="--steps 100 --warmup 10 --workload widedeep_c5"
WL[embedding_mlp_ref]="--steps 200 --warmup 20 --workload embedding_mlp_ref"
cd /tmp && export TMPDIR=/tmp
for w in c5 embedding_mlp_ref; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -o t -- python $R/bench.py ${WL[$w]} $STRICT > $O/${w}_strict.log 2>&1
  grep '^{"metric"' $O/${w}_strict.log | tail -1 > $O/${w}_strict_bench.json
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_strict_kernel_stats.csv
  rm -rf $O/trace_$w
  echo "$w: $(head -2 $O/${w}_strict_kernel_stats.csv | tail -1 | cut -c1-150)"
done
pass() { tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- "$@" > $O/pmc_$tag.log 2>&1; }
CMD="python $R/bench.py --workload widedeep_c5 --steps 20 --warmup 5 $STRICT"
pass c5_fetch FETCH_SIZE -- $CMD
pass c5_write WRITE_SIZE -- $CMD
pass c5_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $CMD
pass c5_sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -- $CMD
cd $R
for w in c5 embedding_mlp_ref; do
  timeout 400 python bench.py ${WL[$w]} $STRICT 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${w}_strict_untraced.json
done
python - <<'PY'
import csv, glob, collections, os, json
summary = {}
for d in sorted(glob.glob('gpurun_out/r04_prof/pmc_c5_*/')):
    tag = os.path.basename(d.rstrip('/'))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if '(anonymous namespace)::' not in k: continue
            short = k.split('(anonymous namespace)::')[1].split('(')[0].split('<')[0]
            if any(s in short for s in ('prep', 'fold', 'absmax', 'split', 'pack', 'build', 'count_small', 'swizzle', 'coef')): continue
            summary.setdefault(tag, {})[short] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
            summary[tag][short]['launches'] = len(next(iter(cs.values())))
json.dump(summary, open('gpurun_out/r04_prof/pmc_summary_c5.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(summary)[:600])
PY
rm -rf $O/pmc_c5_*/
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -2 | tee $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_driver_command.json
python -c "
import json
l=json.loads(open('$O/bench_driver_command.json').read())
print('driver: value %.4g one-batch %.4g frac %.4f hbm %.4f' % (l['value'], l['value_one_batch_per_launch'], l['roofline']['frac'], l['roofline_hbm_resident']['frac']))
for k,w in l['workloads'].items(): print(k, ('%.4g' % w['value'], '%.4f' % w['roofline']['frac']) if 'value' in w else w.get('latency_ms'))"
