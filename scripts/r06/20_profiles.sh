#!/bin/bash
# Round 6 evidence run (GPU box, via gpurun).  Per workload: (1) a STRICT rocprofv3 kernel trace -- one batch per launch, launches in stream
# order; (2) the same command WITHOUT the tracer (the HIP-event time bench.py reports); (3) [r6] the same workload at bench.py's default
# several-batches-per-launch shape (`<w>_many.json`: VERDICT r05 item 8 -- the small reference shapes' multi-batch figure next to the strict one);
# (4) separate --pmc passes (FETCH_SIZE, WRITE_SIZE; SQ counters for the headline kernels).  Configs 4 / 5 cycle bench.py's hbm_cycle working set
# (4 x the Infinity Cache) in every one of these.  scripts/summarize_profiles.py turns gpurun_out/r06_prof into profiles/r06/roofline_table.{md,json}.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_prof
mkdir -p $O
rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6 > $O/rocminfo.txt 2>&1
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
MANY="--cpu-seconds 0 --no-check --hbm-resident 0 --side-workloads= --no-hardware-probe --variants 0"
declare -A WL
WL[c2]="--steps 400 --warmup 40 --input-batches 32"
WL[c2_hbm]="--steps 400 --warmup 40 --big-vocab 8388608 --input-batches 32"
WL[c2_zipf]="--steps 400 --warmup 40 --input-batches 32 --dist zipf"
WL[c2_f32]="--steps 400 --warmup 40 --input-batches 32"
WL[c2_pairs]="--steps 400 --warmup 40 --workload deepfm_c2"
WL[c3]="--steps 60 --warmup 6 --workload din_c3"
WL[c4_v2]="--steps 200 --warmup 20 --workload deepfm_v2_c4"
WL[c4_pairs]="--steps 200 --warmup 20 --workload deepfm_c4"
WL[c5]="--steps 100 --warmup 10 --workload widedeep_c5"
WL[v2_ref]="--steps 400 --warmup 40 --workload deepfm_v2_ref"
WL[ncf_ref]="--steps 400 --warmup 40 --workload neuralcf_ref"
WL[deepfm_ref]="--steps 400 --warmup 40 --workload deepfm_ref"
WL[din_ref]="--steps 100 --warmup 10 --workload din_ref"
WL[embedding_mlp_ref]="--steps 200 --warmup 20 --workload embedding_mlp_ref"
WL[dien_ref]="--steps 100 --warmup 10 --workload dien_ref"
ALL="${WORKLOADS:-c2 c2_hbm c2_zipf c2_f32 c2_pairs c3 c4_v2 c4_pairs c5 v2_ref ncf_ref deepfm_ref din_ref embedding_mlp_ref dien_ref}"
# (WORKLOADS="c5 embedding_mlp_ref": only those workloads again -- after a change to one kernel; the PMC summary then starts from profiles/r06/pmc_summary.json)
has() { case " $ALL " in *" $1 "*) return 0 ;; esac; return 1; }
cd /tmp && export TMPDIR=/tmp
for w in $ALL; do
  export SPRK_V2_HALF=1; [ $w = c2_f32 ] && export SPRK_V2_HALF=0      # (c2_f32: every contraction on f32 MFMA, the exact-fp32 twin)
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -o t -- python $R/bench.py ${WL[$w]} $STRICT > $O/${w}_strict.log 2>&1
  grep '^{"metric"' $O/${w}_strict.log | tail -1 > $O/${w}_strict_bench.json
  f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_strict_kernel_stats.csv
  rm -rf $O/trace_$w
  timeout 400 python $R/bench.py ${WL[$w]} $STRICT 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${w}_strict_untraced.json
  M="${WL[$w]}"; [ $w = c2 ] && M="--steps 400 --warmup 40"                                  # (the headline's own 64 input batches)
  case $w in c2_hbm|c2_zipf|c2_f32) ;; *) timeout 400 python $R/bench.py $M $MANY 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${w}_many.json ;; esac
  echo "$w: $(head -2 $O/${w}_strict_kernel_stats.csv | tail -1 | cut -c1-110)"
done
export SPRK_V2_HALF=1
if has c2; then
# the driver's launch shape (64 batches per launch) under the tracer, EVERY launch of the multi-batch kernel a full one
SPRK_BENCH_SKIP_16=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_drv -o t -- python $R/bench.py --steps 64 --warmup 64 --cpu-seconds 0 --no-check --hbm-resident 0 --side-workloads= --no-hardware-probe > $O/c2_driver.log 2>&1
grep '^{"metric"' $O/c2_driver.log | tail -1 > $O/c2_driver_bench.json
f=$(find $O/trace_drv -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c2_driver_kernel_stats.csv; rm -rf $O/trace_drv
fi
pass() { tag=$1; shift; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- "$@" > $O/pmc_$tag.log 2>&1; }
for w in c2 c2_hbm c2_pairs c3 c4_v2 c4_pairs c5 v2_ref din_ref; do
  has $w || continue
  CMD="python $R/bench.py $(echo ${WL[$w]} | sed -E 's/--steps [0-9]+ --warmup [0-9]+//') --steps 20 --warmup 5 $STRICT"
  pass ${w}_fetch FETCH_SIZE -- $CMD
  pass ${w}_write WRITE_SIZE -- $CMD
done
for w in c2 c2_pairs c3 c5; do
  has $w || continue
  CMD="python $R/bench.py $(echo ${WL[$w]} | sed -E 's/--steps [0-9]+ --warmup [0-9]+//') --steps 20 --warmup 5 $STRICT"
  pass ${w}_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $CMD
  pass ${w}_sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -- $CMD
done
cd $R
python - <<'PY'
import csv, glob, collections, os, json
summary = json.load(open('profiles/r06/pmc_summary.json')) if os.environ.get('WORKLOADS') else {}
for d in sorted(glob.glob('gpurun_out/r06_prof/pmc_*/')):
    tag = os.path.basename(d.rstrip('/'))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if '(anonymous namespace)::' not in k and 'sprk_dev::' not in k: continue
            short = k.replace('(anonymous namespace)::', 'sprk_dev::').split('sprk_dev::')[1].split('(')[0]
            short = short if short.startswith('k_din_fused<') else short.split('<')[0]     # (its TAIL / attention-only forms are two kernels)
            if any(s in short for s in ('prep', 'fold', 'absmax', 'split', 'pack', 'build', 'count_small', 'swizzle', 'coef')): continue
            summary.setdefault(tag, {})[short] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
            summary[tag][short]['launches'] = len(next(iter(cs.values())))
json.dump(summary, open('gpurun_out/r06_prof/pmc_summary.json', 'w'), indent=1, sort_keys=True)
print(len(summary), 'pmc passes summarised')
PY
rm -rf $O/pmc_*/
# the GPU suite and the driver's command on this build
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -6 | tee $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_driver_command.json
ls $O | wc -l
