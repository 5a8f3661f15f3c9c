#!/bin/bash
# Round 5: the strict commands of scripts/r05/20_profiles.sh WITHOUT the tracer -- the HIP-event times bench.py reports, to set
# beside rocprofv3's per-kernel averages (scripts/summarize_profiles.py).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_prof
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads="
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
for w in c2 c2_hbm c2_zipf c2_f32 c2_pairs c3 c4_v2 c4_pairs c5 v2_ref ncf_ref deepfm_ref din_ref embedding_mlp_ref dien_ref; do
  export SPRK_V2_HALF=1; [ $w = c2_f32 ] && export SPRK_V2_HALF=0      # (c2_f32: every contraction on f32 MFMA, the exact-fp32 twin)
  timeout 400 python bench.py ${WL[$w]} $STRICT 2>/dev/null | grep '^{"metric"' | tail -1 > $O/${w}_strict_untraced.json
  python -c "
import json,sys
l=json.loads(open('$O/${w}_strict_untraced.json').read()); r=l['roofline']
print('$w', 'us %.3f' % r['avg_launch_us'], 'frac %.4f' % r['frac'])"
done
# the GPU suite and the driver's command on this build, for profiles/r05
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -6 | tee $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 > $O/bench_driver_command.json
python -c "
import json
l=json.loads(open('$O/bench_driver_command.json').read())
print('driver: value %.4g one-batch %.4g frac %.4f hbm %.4f' % (l['value'], l['value_one_batch_per_launch'], l['roofline']['frac'], l['roofline_hbm_resident']['frac']))
for k,w in l['workloads'].items(): print(k, ('%.4g' % w['value'], '%.4f' % w['roofline']['frac']) if 'value' in w else {a: w[a] for a in list(w)[:6]})
print('cpu', l['cpu_baseline']['value'], l['cpu_baseline'].get('tensorflow'))"
