#!/bin/bash
# GPU box: regenerate the round-2 evidence under gpurun_out/profiles/ (copy what should be judged into profiles/).
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O; cd $R
T=${1:-r02}
tools/ubench/bin/launch_floor 65536 > $O/${T}_launch_floor.json 2>&1
python bench.py > $O/${T}_bench_e2e.json 2> $O/bench_e2e.err
python bench.py --variant indi --no-extras > $O/${T}_bench_indi.json 2> $O/bench_indi.err
for v in e2e indi; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o $v -- python $R/bench.py --variant $v --no-cpu-baseline --no-parity --no-extras > /dev/null 2>&1)
  python tools/rocprof_summary.py /tmp/prof_$v/${v}_results.db > $O/${T}_${v}_kernel_stats.txt 2>&1
done
bash tools/run_pmc.sh > $O/run_pmc.log 2>&1          # HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) -> $O/pmc_summary.json
bash tools/run_pmc_compute.sh $T > $O/run_pmcc.log 2>&1; cp gpurun_out/${T}_pmc_compute.json $O/
python tools/bench_quad3d.py > $O/${T}_quad3d_bench.json 2>/dev/null
(for a in "--obs-len 17" "--obs-len 24" "--minibatch 32768" "--minibatch 65536"; do python tools/bench_ppo_update.py $a 2>/dev/null | tail -1; done) > $O/${T}_ppo_update_bench.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo -o ppo -- python $R/tools/bench_ppo_update.py --iters 100 > /dev/null 2>&1)
python tools/rocprof_summary.py /tmp/prof_ppo/ppo_results.db > $O/${T}_ppo_update_kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_ppo_$c -o p -- python $R/tools/pmc_ppo.py probe > /dev/null 2>&1)
done
python tools/pmc_ppo.py summarise /tmp/pmc_ppo_FETCH_SIZE/p_counter_collection.csv /tmp/pmc_ppo_WRITE_SIZE/p_counter_collection.csv $O/${T}_pmc_ppo_summary.json > /dev/null 2>&1
ls -la $O
# the split form of the PPO gradient (phase A + phase B through HBM scratch) for comparison
(cd /tmp && QR_PPO_SPLIT=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo_split -o ppo -- python $R/tools/bench_ppo_update.py --iters 100 > /dev/null 2>&1)
python tools/rocprof_summary.py /tmp/prof_ppo_split/ppo_results.db > $O/${T}_ppo_update_split_kernel_stats.txt 2>&1
python tools/ppo_phase_timing.py 2>/dev/null | grep -v amdgpu > $O/${T}_ppo_phase_timing.txt
ls -la $O
