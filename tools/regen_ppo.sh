#!/bin/bash
# GPU box: regenerate the PPO-UPDATE evidence under gpurun_out/profiles/ (copy what should be judged into profiles/).
#   QR_COMMIT=<short hash> bash tools/regen_ppo.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O; cd $R
T=${QR_TAG:-r06}
for B in 16384 65536 131072; do python tools/bench_ppo_update.py --obs-len 24 --minibatch $B --iters 200 2>/dev/null | tail -1; done > $O/${T}_ppo_update_bench.json
# rocprofv3 per-kernel averages of the same command (16 384 and 65 536 rows)
for B in 16384 65536; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo_$B -o ppo -- python $R/tools/bench_ppo_update.py --obs-len 24 --minibatch $B --iters 200 > /dev/null 2>&1)
  python tools/rocprof_summary.py /tmp/prof_ppo_$B/ppo_results.db > $O/${T}_ppo_update_${B}_kernel_stats.txt 2>&1
done
# HBM traffic of the two kernels (PMC, separate passes)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_ppo_fetch -o f -- python tools/pmc_ppo.py probe > gpurun_out/pmc_ppo_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_ppo_write -o w -- python tools/pmc_ppo.py probe > gpurun_out/pmc_ppo_w.log 2>&1
python tools/pmc_ppo.py summarise $(find gpurun_out/pmc_ppo_fetch -name 'f_counter_collection.csv') $(find gpurun_out/pmc_ppo_write -name 'w_counter_collection.csv') $O/${T}_pmc_ppo_summary.json > $O/${T}_pmc_ppo.log 2>&1
# wall-clock split of the two launches and the in-wave phase stamps
python tools/ppo_launch_timing.py 24 16384 > $O/${T}_ppo_launch_timing.txt 2>&1
ls -la $O
