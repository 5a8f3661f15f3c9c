#!/bin/bash
# Round-2 evidence run on the GPU box: bench lines of every workload, rocprofv3 --kernel-trace --stats of the same commands,
# PMC passes (separate runs, no tracing domains besides --kernel-trace) -> gpurun_out/r02/ (copied to profiles/r02/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6 > $O/rocminfo.txt 2>&1
b() { out=$1; shift; timeout 900 "$@" > $O/$out.json 2> $O/$out.err; tail -1 $O/$out.json | cut -c1-160; }
echo "=== bench lines"
b bench_c2_driver python bench.py --gpus 1 --steps 20 --warmup 5
b bench_c2 python bench.py --cpu-seconds 0
b bench_c2_lb1 python bench.py --cpu-seconds 0 --launch-batches 1 --hbm-resident 0
b bench_c2_strict python bench.py --cpu-seconds 0 --launch-batches 1 --overlap-streams 0 --hbm-resident 0
b bench_c2_zipf python bench.py --cpu-seconds 0 --dist zipf --hbm-resident 0
b bench_c2_f32 env SPRK_V2_HALF=0 python bench.py --cpu-seconds 0 --hbm-resident 0
b bench_c2_rows env SPRK_V2_ROWS=1 python bench.py --cpu-seconds 0 --hbm-resident 0
b bench_c2_interp env SPRK_FORCE_INTERPRETER=1 python bench.py --cpu-seconds 0 --hbm-resident 0 --steps 200 --warmup 20
b bench_c2_pairs python bench.py --workload deepfm_c2 --cpu-seconds 0
b bench_c2_pairs_f32 env SPRK_DYN_F16=0 python bench.py --workload deepfm_c2 --cpu-seconds 0
b bench_c3 python bench.py --workload din_c3 --steps 320 --warmup 32 --cpu-seconds 6
b bench_c3_wpb4 env SPRK_DIN_WPB=4 python bench.py --workload din_c3 --steps 320 --warmup 32 --cpu-seconds 0
b bench_c4_v2 python bench.py --workload deepfm_v2_c4 --steps 400 --warmup 40 --cpu-seconds 0
b bench_c4_pairs python bench.py --workload deepfm_c4 --steps 200 --warmup 20 --cpu-seconds 0
b bench_c5 python bench.py --workload widedeep_c5 --steps 200 --warmup 20 --cpu-seconds 0
b bench_c5_chain env SPRK_MLP_ROWS=0 python bench.py --workload widedeep_c5 --steps 200 --warmup 20 --cpu-seconds 0
b bench_v2_ref python bench.py --workload deepfm_v2_ref --steps 400 --warmup 40 --cpu-seconds 0
b bench_v2_ref_interp env SPRK_FORCE_INTERPRETER=1 python bench.py --workload deepfm_v2_ref --steps 100 --warmup 10 --cpu-seconds 0
b bench_ncf_ref python bench.py --workload neuralcf_ref --steps 400 --warmup 40 --cpu-seconds 0
b bench_ncf_ref_interp env SPRK_NCF_CHAIN=0 python bench.py --workload neuralcf_ref --steps 400 --warmup 40 --cpu-seconds 0
b bench_c4_pairs_interp env SPRK_V1_CHAIN=0 python bench.py --workload deepfm_c4 --steps 100 --warmup 10 --cpu-seconds 0
b bench_c2_forced_collective_sprk env SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 python bench.py --cpu-seconds 0 --hbm-resident 0 --collective sprk
b bench_c2_forced_collective_torch env SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 python bench.py --cpu-seconds 0 --hbm-resident 0 --collective torch
b bench_c2_forced_collective_peer env SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 python bench.py --cpu-seconds 0 --hbm-resident 0 --collective peer
b bench_c2_gloo2_peer python bench.py --gpus 2 --backend gloo --collective peer --steps 20 --warmup 5 --cpu-seconds 0 --hbm-resident 0 --min-region-ms 1 --regions 1 --settle-ms 0
b bench_c3_attn_per_batch env SPRK_DIN_ATTN_MB=0 python bench.py --workload din_c3 --steps 320 --warmup 32 --cpu-seconds 0
b bench_c3_hot python bench.py --workload din_c3 --steps 320 --warmup 32 --cpu-seconds 0 --dist hot
b bench_c2_pairs_strict python bench.py --workload deepfm_c2 --cpu-seconds 0 --launch-batches 1 --overlap-streams 0
b bench_c2_pairs_uploaded_tables env SPRK_V1_ROWTAB=0 python bench.py --workload deepfm_c2 --cpu-seconds 0
timeout 600 python scripts/bench_ingest.py --rows 20000000 --threads 1,128 --device > $O/bench_ingest_20m.json 2> $O/bench_ingest_20m.err; cut -c1-300 $O/bench_ingest_20m.json
b bench_c2_gloo2 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --cpu-seconds 0 --hbm-resident 0 --min-region-ms 1 --regions 1 --settle-ms 0
[ -x scripts/ubench/launch_floor ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/launch_floor.hip -o scripts/ubench/launch_floor
timeout 120 scripts/ubench/launch_floor > $O/ubench_launch_floor.log 2>&1; tail -20 $O/ubench_launch_floor.log
timeout 900 python -m pytest tests -m gpu -q --durations=5 2>&1 | grep -v "^NCCL\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|^$" | tail -12 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
echo "=== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o $tag -- "$@" > $O/prof_$tag.log 2>&1; f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv && head -4 $f | cut -c1-160; }
Q="--cpu-seconds 0 --no-check --hbm-resident 0 --regions 2"
prof c2_driver python $R/bench.py --gpus 1 --steps 20 --warmup 5 $Q
prof c2_strict python $R/bench.py --steps 400 --warmup 40 --launch-batches 1 --overlap-streams 0 $Q
prof c2_hbm_resident python $R/bench.py --steps 400 --warmup 40 --launch-batches 1 --overlap-streams 0 --big-vocab 8388608 $Q
prof c2_pairs_strict python $R/bench.py --workload deepfm_c2 --steps 400 --warmup 40 --launch-batches 1 --overlap-streams 0 $Q
prof c3 python $R/bench.py --workload din_c3 --steps 64 --warmup 8 $Q
prof c4_v2_strict python $R/bench.py --workload deepfm_v2_c4 --steps 400 --warmup 40 --launch-batches 1 --overlap-streams 0 $Q
prof c5 python $R/bench.py --workload widedeep_c5 --steps 100 --warmup 10 $Q
prof c4_pairs_strict python $R/bench.py --workload deepfm_c4 --steps 200 --warmup 20 --launch-batches 1 --overlap-streams 0 $Q
prof v2_ref_strict python $R/bench.py --workload deepfm_v2_ref --steps 400 --warmup 40 --launch-batches 1 --overlap-streams 0 $Q
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ingest -o ingest -- python $R/scripts/bench_ingest.py --rows 20000000 --threads 128 --device > $O/prof_ingest.log 2>&1; f=$(find $O/prof_ingest -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/ingest_kernel_stats.csv
prof ncf_ref_strict python $R/bench.py --workload neuralcf_ref --steps 400 --warmup 40 --launch-batches 1 --overlap-streams 0 $Q
echo "=== PMC passes"
pass() { # tag name counters -- command...
  tag=$1; name=$2; shift 2; ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $O/pmc_${tag}_$name -o p -- "$@" > $O/pmc_${tag}_$name.log 2>&1
}
P="--steps 20 --warmup 5 --cpu-seconds 0 --no-check --hbm-resident 0 --regions 1 --min-region-ms 1 --settle-ms 0 --launch-batches 1 --overlap-streams 0"
for t in c2 c2hbm pairs c3 c5 v2ref c4pairs cal; do
  case $t in
    c2) CMD="python $R/bench.py $P";;
    c2hbm) CMD="python $R/bench.py $P --big-vocab 8388608";;
    pairs) CMD="python $R/bench.py $P --workload deepfm_c2";;
    c3) CMD="python $R/bench.py $P --workload din_c3";;
    c5) CMD="python $R/bench.py $P --workload widedeep_c5";;
    v2ref) CMD="python $R/bench.py $P --workload deepfm_v2_ref";;
    c4pairs) CMD="python $R/bench.py $P --workload deepfm_c4";;
    cal) CMD="python $R/scripts/pmc_calib.py";;
  esac
  pass $t fetch FETCH_SIZE -- $CMD
  pass $t write WRITE_SIZE TCC_REQ_sum -- $CMD
  if [ $t != cal ]; then
    pass $t sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- $CMD
    pass $t sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT -- $CMD
  fi
done
cd $R
python - <<'PY'
import csv, glob, collections, os, json
O = 'gpurun_out/r02'
summary = {}
for d in sorted(glob.glob(O + '/pmc_*/')):
    tag = os.path.basename(d.rstrip('/'))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in agg.items():
            if any(x in k for x in ('rocclr', 'at::', 'prep', 'fold', 'absmax', 'split', 'pack', 'build', 'elementwise', 'Cijk')):
                continue
            short = k.split('(anonymous namespace)::')[1].split('(')[0].split('<')[0] if '(anonymous namespace)::' in k else k[:40]
            if 'many' in k.split('(')[0]:
                short += '_many'
            summary.setdefault(tag, {})[short] = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
            summary[tag][short]['launches'] = len(next(iter(cs.values())))
json.dump(summary, open(O + '/pmc_summary.json', 'w'), indent=1, sort_keys=True)
for tag, ks in summary.items():
    for k, v in ks.items():
        print(tag, k, v)
PY
grep -h "calib D" $O/pmc_cal_fetch.log
# keep the scratch small: only summaries travel back
find $O -name "*.db" -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +2M -delete 2>/dev/null
du -sh $O
