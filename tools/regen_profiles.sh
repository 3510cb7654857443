#!/bin/bash
# GPU box: regenerate the evidence under gpurun_out/profiles/ (copy what should be judged into profiles/).
#   bench JSON lines (configs 2, 3), rocprofv3 kernel stats of the same commands, predecessor-env and PPO-update measurements
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O; cd $R
python bench.py > $O/r01_bench_e2e.json 2> $O/bench_e2e.err
python bench.py --variant indi > $O/r01_bench_indi.json 2> $O/bench_indi.err
for v in e2e indi; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o $v -- python $R/bench.py --variant $v --no-cpu-baseline --no-parity > /dev/null 2>&1)
  python tools/rocprof_summary.py /tmp/prof_$v/${v}_results.db > $O/r01_${v}_kernel_stats.txt 2>&1
done
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_q3 -o q3 -- python $R/tools/bench_quad3d.py --steps 300 > /dev/null 2>&1)
python tools/rocprof_summary.py /tmp/prof_q3/q3_results.db > $O/r01_quad3d_kernel_stats.txt 2>&1
python tools/bench_quad3d.py > $O/r01_quad3d_bench.json 2>/dev/null
python tools/bench_quad3d.py --envs 1048576 --steps 100 > $O/r01_quad3d_bench_1Mi.json 2>/dev/null
(for a in "--obs-len 17" "--obs-len 24" "--minibatch 32768" "--minibatch 65536"; do python tools/bench_ppo_update.py $a 2>/dev/null | tail -1; done) > $O/r01_ppo_update_bench.json
python tools/ppo_phase_timing.py 2>/dev/null | tail -17 > $O/r01_ppo_phase_timing.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo -o ppo -- python $R/tools/train_ppo.py --variant indi --steps 6.3e7 --n-steps 32 --epochs 10 --minibatches 128 --lr 3e-4 --target-kl 1e9 --fused --native-update --gamma 0.99 > /dev/null 2>&1)
python tools/rocprof_summary.py /tmp/prof_ppo/ppo_results.db > $O/r01_ppo_native_kernel_stats.txt 2>&1
ls -la $O
