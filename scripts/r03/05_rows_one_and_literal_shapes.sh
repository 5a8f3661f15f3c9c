#!/bin/bash
# Round 3, GPU call 5: k_rows_chain1 (one task per wave) A/B on the reference-literal DeepFM_v2 / NeuralCF shapes, the other
# reference-literal shapes' first bench lines (DeepFM.py, DIN.py, EmbeddingMLP.py), the whole GPU suite, the driver line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_05
mkdir -p $O
echo "=== tests"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(HIP|ROCm|Hostname|Librccl|RCCL|$)" | tail -12 | tee $O/pytest_gpu.log
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print(sys.argv[1].split('/')[-1], l['config']['kernel'], 'value %.3g' % l['value'], 'us/step %.3f' % (l['ms_per_step'] * 1e3),
          'strict us %.3f frac %.4f' % (l['roofline'].get('step_us_all_kernels', l['roofline']['avg_launch_us']), l['roofline']['frac']), 'err', l['config']['oracle_check_max_abs_err'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
b v2_ref_one python bench.py --workload deepfm_v2_ref --cpu-seconds 0
b v2_ref_loop SPRK_ROWS_ONE=0 python bench.py --workload deepfm_v2_ref --cpu-seconds 0
b ncf_ref_one python bench.py --workload neuralcf_ref --cpu-seconds 0
b ncf_ref_loop SPRK_ROWS_ONE=0 python bench.py --workload neuralcf_ref --cpu-seconds 0
b deepfm_ref python bench.py --workload deepfm_ref --cpu-seconds 0
b din_ref python bench.py --workload din_ref --cpu-seconds 0 --steps 320 --warmup 32
b embedding_mlp_ref python bench.py --workload embedding_mlp_ref --cpu-seconds 0
for f in deepfm_ref din_ref embedding_mlp_ref; do tail -2 $O/$f.err; done
echo "=== driver line"; t0=$(date +%s); timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "wall $(( $(date +%s) - t0 )) s"
tail -1 $O/bench_driver.json | python -c "
import sys, json
l = json.loads(sys.stdin.readline())
print('value', l['value'], 'one-batch', l['value_one_batch_per_launch'], 'frac', l['roofline']['frac'])
print('hbm_resident', {k: l['roofline_hbm_resident'][k] for k in ('frac','avg_launch_us','working_set_mb','frac_16_batches_per_launch')})
for k, w in l.get('workloads', {}).items():
    print(k, w['kernel'], w['value'], w['ms_per_step'], w['roofline']['frac'], w['roofline']['avg_launch_us'], w['oracle_check_max_abs_err'])
"
