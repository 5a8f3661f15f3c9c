#!/usr/bin/env python3
"""Reads the stamps a -DSPRK_DF_XP build of k_din_fused leaves (SPRK_DF_XP=1024, SPRK_DF_TS_FILE): per wave the 100 MHz clock at kernel
entry, with the ids in LDS, with the second round trip landed, at slot-loop entry (behind the barrier), loop exit, after fc0, after fc1 and at exit (BASELINE config 3: 2048 waves)."""
import sys, json, numpy as np
ts = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)[:2048].astype(np.int64)
l = json.loads(open(sys.argv[2]).read())
print('bench: step %.2f us (stamped build)' % l['roofline']['step_us_all_kernels'])
t0 = ts[:, 0].min()
names = ['entry', 'ids in LDS', 'trip 2 landed', 'loop entry', 'loop exit', 'fc0 done', 'fc1 done', 'exit']
rel = (ts[:, [0, 3, 7, 1, 2, 4, 5, 6]] - t0) * 0.01     # us
print('%-12s %8s %8s %8s %8s %8s' % ('stamp', 'min', 'p10', 'median', 'p90', 'max'))
for k, n in enumerate(names):
    c = rel[:, k]
    print('%-12s %8.2f %8.2f %8.2f %8.2f %8.2f' % (n, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
print('phase durations per wave (us): median [p10 .. p90]')
for k in range(1, 8):
    d = rel[:, k] - rel[:, k - 1]
    print('  %-12s -> %-12s %6.2f [%6.2f .. %6.2f]' % (names[k - 1], names[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
wg = rel.reshape(256, 8, 8)
print('per workgroup: spread of loop exit across its 8 waves, median %.2f us; of exit %.2f us' % (np.median(wg[:, :, 4].max(1) - wg[:, :, 4].min(1)), np.median(wg[:, :, 7].max(1) - wg[:, :, 7].min(1))))
