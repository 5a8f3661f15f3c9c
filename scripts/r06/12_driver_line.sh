#!/bin/bash
# Round 6: the driver's command (python bench.py, no flags): wall time and the kept keys
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06_20}
mkdir -p $O
t0=$(date +%s.%N)
python bench.py > $O/bench_default.json 2> $O/bench_default.err
t1=$(date +%s.%N)
echo "wall $(python -c "print(round($t1-$t0,1))") s"
tail -1 $O/bench_default.json | python -c "
import sys,json
l=json.loads(sys.stdin.read())
print('value %.4g %s, ms_per_step %.5f' % (l['value'], l['unit'], l['ms_per_step']))
r=l['roofline']; print({k: (round(v,4) if isinstance(v,float) else v) for k,v in r.items() if not isinstance(v,(dict,str)) })
c=l['config']; print({k: (round(v,4) if isinstance(v,float) else v) for k,v in c.items() if any(s in k for s in ('_us','_frac','_per_s','working_set','_GBps','cycled','p50'))})
print(l['cpu_baseline'])
"
tail -3 $O/bench_default.err
