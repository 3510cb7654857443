#!/bin/bash
# GPU box: regenerate the ENV-KERNEL evidence under gpurun_out/profiles/ (copy what should be judged into profiles/).
#   QR_COMMIT=<short hash> bash tools/regen_profiles.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O; cd $R
T=${QR_TAG:-r06}
# counters first: bench.py looks the traffic of its kernels up in profiles/r05_pmc_*.json
QR_TAG=$T bash tools/run_pmc.sh > $O/${T}_run_pmc.log 2>&1; cp $O/${T}_pmc_traffic.json profiles/ 2>/dev/null
QR_PMC_ONLY_ENV=1 bash tools/run_pmc_compute.sh ${T} > $O/${T}_run_pmc_compute.log 2>&1; cp gpurun_out/${T}_pmc_compute.json $O/; cp gpurun_out/${T}_pmc_compute.json profiles/
QR_PMC_ENVS=1048576 QR_PMC_ONLY_ENV=1 bash tools/run_pmc_compute.sh ${T}_n1Mi > $O/${T}_run_pmc_compute_1Mi.log 2>&1; cp gpurun_out/${T}_n1Mi_pmc_compute.json $O/
python bench.py --steps 20 --warmup 5 > $O/${T}_bench_e2e_k20.json 2> $O/${T}_bench_e2e_k20.full.json
python bench.py > $O/${T}_bench_e2e.json 2> $O/${T}_bench_e2e.full.json
python bench.py --variant indi --no-extras > $O/${T}_bench_indi.json 2> $O/${T}_bench_indi.full.json
# rocprofv3 per-kernel averages of the SAME commands the lines above come from (driver's K = 20 and the default K = 1000)
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_k20 -o k20 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras > /dev/null 2>&1)
python tools/rocprof_summary.py /tmp/prof_k20/k20_results.db > $O/${T}_e2e_k20_kernel_stats.txt 2>&1
for v in e2e indi; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o $v -- python $R/bench.py --variant $v --no-cpu-baseline --no-parity --no-extras > /dev/null 2>&1)
  python tools/rocprof_summary.py /tmp/prof_$v/${v}_results.db > $O/${T}_${v}_kernel_stats.txt 2>&1
done
bash tools/sweep_envs.sh > $O/${T}_sweep_envs.txt 2>&1
(for n in 131072 262144 1048576 4194304; do python tools/fused_probe.py e2e $n 100; done; python tools/fused_probe.py indi 1048576 100; QR_ROLLOUT_FORM=general python tools/fused_probe.py e2e 1048576 100) 2>&1 | grep "us/step" > $O/${T}_fused_probe.txt
(python tools/step_probe.py e2e; python tools/step_probe.py indi) 2>&1 | grep -v amdgpu.ids > $O/${T}_step_probe.txt
ls -la $O
