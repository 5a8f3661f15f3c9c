#!/bin/bash
# Round 3: the N>1 loop of bench.py on the one-GPU box -- RCCL at world 1 (torch / C ABI / peer writes), and two gloo-launched
# ranks sharing the device (functional) -- after the move to sprk_forward_many_opts.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_12
mkdir -p $O
b() { out=$1; shift; timeout 300 env "$@" 2>$O/$out.err | grep '^{"metric"' | tail -1 > $O/$out.json; python - $O/$out.json <<'PY'
import sys, json
try:
    l = json.loads(open(sys.argv[1]).read())
    print(sys.argv[1].split('/')[-1], 'n_gpus', l['n_gpus'], 'value %.4g' % l['value'], 'us/step %.3f' % (l['ms_per_step'] * 1e3), l['config'].get('collective'), 'err', l['config']['oracle_check_max_abs_err'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
b forced_torch SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 python bench.py --cpu-seconds 0
b forced_sprk SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 python bench.py --cpu-seconds 0 --collective sprk
b forced_peer SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 python bench.py --cpu-seconds 0 --collective peer
b gloo2 python bench.py --gpus 2 --backend gloo --cpu-seconds 0 --steps 64 --warmup 8 --regions 2 --min-region-ms 5
b gloo2_peer python bench.py --gpus 2 --backend gloo --collective peer --cpu-seconds 0
b din_forced SPRK_BENCH_FORCE_DIST=1 SPRK_FORCE_COLLECTIVE=1 python bench.py --cpu-seconds 0 --workload din_c3 --steps 320 --warmup 32
for f in forced_torch forced_sprk forced_peer gloo2 gloo2_peer din_forced; do tail -2 $O/$f.err | cut -c1-200; done
