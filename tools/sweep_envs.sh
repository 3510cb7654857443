#!/bin/bash
# env-count sweep of both paths (GPU box): bash tools/sweep_envs.sh
# fused = qr_step_many (roofline on the bytes that kernel moves: 118 / 90 B per env-step), per-step = qr_step_launches (285 / 209 B)
echo "# bash tools/sweep_envs.sh on MI355X (bench.py --envs N --no-extras); frac = of 8 000 GB/s"
for v in e2e indi; do for n in 4096 16384 65536 262144 1048576 4194304; do
  k=1000; if [ $n -ge 1048576 ]; then k=100; fi
  python bench.py --variant $v --envs $n --steps $k --warmup 20 --no-cpu-baseline --no-parity --no-extras --repeats 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v %8d  fused %7.2f us/step %6.2f G/s %5.0f GB/s frac %.2f | per-step %7.2f us/step %6.2f G/s frac %.2f' % ($n, d['ms_per_step']*1e3, d['value']/1e9, r['achieved'], r['frac'], r['per_step_kernel_us'], r['per_step_value']/1e9, r['per_step_frac']))"
done; done
