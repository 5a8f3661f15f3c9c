#!/usr/bin/env python3
"""Reads the stamps a -DSPRK_DF_XP build of k_deepfm_v2_joint1 leaves (SPRK_V2J1_TS_FILE): per wave (= task) the 100 MHz clock at
kernel entry, with its ids staged, with its gathers requested, behind the barrier, after phase A (everything that needs no rows),
with its rows landed, at exit.  BASELINE config 2: 4096 waves."""
import sys, numpy as np
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ts = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)[:n].astype(np.int64)
names = ['entry', 'ids staged', 'gathers out', 'past barrier', 'phase A done', 'rows landed', 'exit']
t0 = ts[:, 0].min()
rel = (ts[:, :7] - t0) * 0.01
print('%-13s %7s %7s %7s %7s %7s' % ('stamp', 'min', 'p10', 'median', 'p90', 'max'))
for k, nm in enumerate(names):
    c = rel[:, k]
    print('%-13s %7.2f %7.2f %7.2f %7.2f %7.2f' % (nm, c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
print('phase durations per wave (us): median [p10 .. p90]')
for k in range(1, 7):
    d = rel[:, k] - rel[:, k - 1]
    print('  %-13s -> %-13s %6.2f [%6.2f .. %6.2f]' % (names[k - 1], names[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
