#!/bin/bash
# Round 4: the driver's command with the new blocks (configs 4 / 5, the 800-candidate REST request, the timed region's bound label, the
# 16-batch figure) and its wall time; --scaling strong at N = 1 and with the collective forced on the one GPU (torch / sprk / peer).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_08
mkdir -p $O
t0=$(date +%s.%N)
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench_driver_command.json
t1=$(date +%s.%N)
echo "driver command wall: $(echo "$t1 - $t0" | bc) s"
tail -3 $O/bench.err
python - $O/bench_driver_command.json <<'PY'
import sys, json
l = json.loads(open(sys.argv[1]).read())
print({k: l.get(k) for k in ("value", "ms_per_step", "value_one_batch_per_launch", "value_16_batches_per_launch", "scaling")})
print('roofline', {k: l['roofline'].get(k) for k in ('kernel', 'frac', 'avg_launch_us')})
print('timed_region', {k: l['roofline_timed_region'].get(k) for k in ('bound', 'frac', 'launches', 'avg_launch_us')})
print('hbm_resident', {k: l['roofline_hbm_resident'].get(k) for k in ('frac', 'avg_launch_us', 'frac_16_batches_per_launch')})
for k, w in l.get('workloads', {}).items():
    if 'latency_ms' in w: print(k, w['latency_ms'], 'predict ms %.3f' % w['model_predict_ms'], 'forward us %.2f' % w['forward_launch_us'], 'req/s %.0f' % w['requests_per_sec_one_client'])
    else: print(k, {x: w.get(x) for x in ('value', 'ms_per_step', 'value_one_batch_per_launch', 'kernel', 'oracle_check_max_abs_err')}, 'frac %.3f' % w['roofline']['frac'], w['roofline'].get('avg_launch_us'))
PY
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print('%-26s' % sys.argv[1].split('/')[-1], l['scaling'], 'value %.4g' % l['value'], 'us/step %.2f' % (l['ms_per_step'] * 1e3), 'global batch', l['config']['global_batch'],
          'one request: %.1f us' % l['strong_one_global_batch']['latency_us'], l['strong_one_global_batch']['gathered_scores_oracle_check_max_abs_err'], l['config']['collective'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
COMMON="--steps 200 --warmup 20 --cpu-seconds 0 --side-workloads= --hbm-resident 0 --no-hardware-probe --scaling strong"
b strong_n1 python bench.py $COMMON
b strong_forced_torch SPRK_BENCH_FORCE_DIST=1 python bench.py $COMMON --collective torch
b strong_forced_sprk SPRK_BENCH_FORCE_DIST=1 python bench.py $COMMON --collective sprk
b strong_forced_peer SPRK_BENCH_FORCE_DIST=1 python bench.py $COMMON --collective peer
b strong_din_n1 python bench.py $COMMON --workload din_c3
