#!/bin/bash
# Round 4: k_din_fused's timeline on BASELINE config 3 (library built with -DSPRK_DF_XP, SPRK_DF_XP=1024): every wave stamps the
# 100 MHz clock at kernel entry, slot-loop entry, loop exit, after the cross-wave combine, after fc0, after fc1 and at exit.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_15
mkdir -p $O
STRICT="--cpu-seconds 0 --no-check --launch-batches 1 --overlap-streams 0 --hbm-resident 0 --side-workloads= --no-hardware-probe"
SPRK_DF_XP=1024 SPRK_DF_TS_FILE=$O/ts.bin timeout 200 python bench.py --workload din_c3 --steps 40 --warmup 8 $STRICT 2>$O/ts.err | tail -1 > $O/ts.json
python - $O/ts.bin $O/ts.json <<'PY' | tee $O/timeline.txt
import sys, json, numpy as np
ts = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)[:2048].astype(np.int64)
l = json.loads(open(sys.argv[2]).read())
print('bench: step %.2f us' % l['roofline']['step_us_all_kernels'])
t0 = ts[:, 0].min()
names = ['entry', 'loop entry', 'loop exit', 'combined', 'fc0 done', 'fc1 done', 'exit']
rel = (ts[:, :7] - t0) * 0.01     # us
print('%-12s %8s %8s %8s %8s %8s' % ('stamp', 'min', 'p10', 'median', 'p90', 'max'))
for k, n in enumerate(names):
    c = rel[:, k]
    print('%-12s %8.2f %8.2f %8.2f %8.2f %8.2f' % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
print('phase durations per wave (us): median [p10 .. p90]')
for k in range(1, 7):
    d = rel[:, k] - rel[:, k - 1]
    print('  %-12s -> %-12s %6.2f [%6.2f .. %6.2f]' % (names[k - 1], names[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
wg = rel.reshape(256, 8, 7)
print('per workgroup: spread of loop exit across its 8 waves, median %.2f us; of exit %.2f us' % (np.median(wg[:, :, 2].max(1) - wg[:, :, 2].min(1)), np.median(wg[:, :, 6].max(1) - wg[:, :, 6].min(1))))
PY
