#!/bin/bash
# env-count sweep of both paths (GPU box): bash tools/sweep_envs.sh
for v in e2e indi; do for n in 4096 16384 65536 262144 1048576 4194304; do
  k=1000; if [ $n -ge 1048576 ]; then k=100; fi
  python bench.py --variant $v --envs $n --steps $k --warmup 20 --no-cpu-baseline --no-parity --repeats 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['per_step_launch']
print('$v %8d  fused %7.2f us/step %6.2f G/s alg %5.0f GB/s | per-step %7.2f us/step %6.2f G/s alg %5.0f GB/s' % ($n, d['ms_per_step']*1e3, d['value']/1e9, d['roofline']['achieved'], p['ms_per_step']*1e3, p['value']/1e9, p['roofline']['achieved']))"
done; done
