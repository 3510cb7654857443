#!/bin/bash
# GPU box: regenerate the round-3 evidence under gpurun_out/profiles/ (copy what should be judged into profiles/).
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O; cd $R
T=r03
python bench.py --steps 20 --warmup 5 > $O/${T}_bench_e2e_k20.json 2> $O/${T}_bench_e2e_k20.full.json
python bench.py > $O/${T}_bench_e2e.json 2> $O/${T}_bench_e2e.full.json
python bench.py --variant indi --no-extras > $O/${T}_bench_indi.json 2> $O/${T}_bench_indi.full.json
for v in e2e indi; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o $v -- python $R/bench.py --variant $v --no-cpu-baseline --no-parity --no-extras > /dev/null 2>&1)
  python tools/rocprof_summary.py /tmp/prof_$v/${v}_results.db > $O/${T}_${v}_kernel_stats.txt 2>&1
done
(for a in "--obs-len 17" "--obs-len 24" "--obs-len 24 --minibatch 32768" "--obs-len 24 --minibatch 65536"; do python tools/bench_ppo_update.py $a 2>/dev/null | tail -1; done) > $O/${T}_ppo_update_bench.json
(for a in "--obs-len 24" "--obs-len 24 --minibatch 65536"; do QR_PPO_GRAD4=1 python tools/bench_ppo_update.py $a 2>/dev/null | tail -1; done) > $O/${T}_ppo_update_bench_grad4.json
(for a in "--obs-len 24" "--obs-len 24 --minibatch 65536"; do QR_PPO_PARTIAL=f32 python tools/bench_ppo_update.py $a 2>/dev/null | tail -1; done) > $O/${T}_ppo_update_bench_f32partials.json
(python tools/ppo_launch_timing.py 24 16384; QR_PPO_PARTIAL=f32 python tools/ppo_launch_timing.py 24 16384) 2>/dev/null | grep -v amdgpu > $O/${T}_ppo_launch_timing.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo -o ppo -- python $R/tools/bench_ppo_update.py --obs-len 24 --iters 100 > /dev/null 2>&1)
python tools/rocprof_summary.py /tmp/prof_ppo/ppo_results.db > $O/${T}_ppo_update_kernel_stats.txt 2>&1
(cd /tmp && QR_PPO_GRAD4=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo4 -o ppo -- python $R/tools/bench_ppo_update.py --obs-len 24 --iters 100 > /dev/null 2>&1)
python tools/rocprof_summary.py /tmp/prof_ppo4/ppo_results.db > $O/${T}_ppo_update_grad4_kernel_stats.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo64 -o ppo -- python $R/tools/bench_ppo_update.py --obs-len 24 --minibatch 65536 --iters 60 > /dev/null 2>&1)
python tools/rocprof_summary.py /tmp/prof_ppo64/ppo_results.db > $O/${T}_ppo_update_65536_kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_ppo_$c -o p -- python $R/tools/pmc_ppo.py probe > /dev/null 2>&1)
done
python tools/pmc_ppo.py summarise /tmp/pmc_ppo_FETCH_SIZE/p_counter_collection.csv /tmp/pmc_ppo_WRITE_SIZE/p_counter_collection.csv $O/${T}_pmc_ppo_summary.json > /dev/null 2>&1
(QR_TICK_NODRAIN=1 python tools/ppo_phase_timing.py 2>/dev/null | grep -v amdgpu) > $O/${T}_ppo_phase_timing.txt
(QR_PPO_GRAD4=1 python tools/ppo_phase_timing.py 2>/dev/null | grep -v amdgpu) > $O/${T}_ppo_phase_timing_grad4.txt
ls -la $O
