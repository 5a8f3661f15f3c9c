#!/usr/bin/env python3
"""Reads the stamps a -DSPRK_DF_XP build of k_mlp_rows leaves (SPRK_MR_TS_FILE): per wave the 100 MHz clock at kernel entry, behind the
workgroup's meeting, with the first gather requested, and per trip: rows summed into the accumulators | next gather requested | first layer done
(genre rows + numerics + ReLU) | second layer done | scores stored.  BASELINE config 5: 2048 waves x 4 trips."""
import sys, json, numpy as np
SLOTS = 32
ts = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, SLOTS)[:2048].astype(np.int64)
l = json.loads(open(sys.argv[2]).read())
print('bench: strict launch %.2f us (stamped build)' % l['roofline']['avg_launch_us'])
t0 = ts[:, 0].min()
rel = (ts - t0) * 0.01     # us
ntrips = 0
while 7 + 6 * ntrips < SLOTS and (ts[:, 7 + 6 * ntrips] > 0).mean() > 0.5:
    ntrips += 1
names = ['entry', 'behind the meeting', 'first gather out']
for t in range(ntrips):
    names += ['trip %d: rows summed' % t, 'trip %d: next gather out' % t, 'trip %d: first layer done' % t, 'trip %d: second layer done' % t, 'trip %d: stored' % t]
cols = [0, 1, 2] + [3 + 6 * t + k for t in range(ntrips) for k in range(5)]
print('%-28s %8s %8s %8s %8s %8s   %s' % ('stamp', 'min', 'p10', 'median', 'p90', 'max', 'since the stamp before: median [p10 .. p90]'))
prev = None
for n, c in zip(names, cols):
    v = rel[:, c]
    line = '%-28s %8.2f %8.2f %8.2f %8.2f %8.2f' % (n, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
    if prev is not None:
        d = v - rel[:, prev]
        line += '   %6.2f [%6.2f .. %6.2f]' % (np.median(d), np.percentile(d, 10), np.percentile(d, 90))
    print(line)
    prev = c
print('per trip (median over waves, us): wait for rows | gather issue | first layer | second layer | head + store')
for t in range(ntrips):
    b = 3 + 6 * t
    before = rel[:, b - 2] if t else rel[:, 2]
    seg = [rel[:, b] - before, rel[:, b + 1] - rel[:, b], rel[:, b + 2] - rel[:, b + 1], rel[:, b + 3] - rel[:, b + 2], rel[:, b + 4] - rel[:, b + 3]]
    print('  trip %d: ' % t + ' | '.join('%5.2f' % np.median(x) for x in seg) + '   = %.2f' % sum(np.median(x) for x in seg))
if (ts[:, 27] > 0).mean() > 0.5:
    print('the first gather (us, median [p10 .. p90]): meeting -> numerics operands requested | -> ids read from the slot | -> big rows requested | -> cross row requested | -> next ids requested')
    seq = [1, 27, 28, 29, 30, 2]
    print('   ' + ' | '.join('%5.2f [%5.2f .. %5.2f]' % (np.median(d), np.percentile(d, 10), np.percentile(d, 90)) for d in (rel[:, seq[i + 1]] - rel[:, seq[i]] for i in range(5))))
# the two waves of a SIMD: wave w of a workgroup sits on SIMD w % 4; the older is the lower wave id
wg = rel.reshape(-1, 8, SLOTS)
last = 7 + 6 * (ntrips - 1)
print('exit, waves 0-3 (older on their SIMD) median %.2f us, waves 4-7 %.2f us; launch ends %.2f us after the first entry' % (
    np.median(wg[:, :4, last]), np.median(wg[:, 4:, last]), rel[:, last].max()))
