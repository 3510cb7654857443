#!/bin/bash
# GPU box: regenerate the round-4 ENV-KERNEL evidence under gpurun_out/profiles/ (copy what should be judged into profiles/).
#   QR_COMMIT=<short hash> bash tools/regen_profiles_r04.sh
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O; cd $R
T=r04
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
bash tools/run_pmc.sh > $O/${T}_run_pmc.log 2>&1
QR_PMC_ONLY_ENV=1 bash tools/run_pmc_compute.sh ${T} > $O/${T}_run_pmc_compute.log 2>&1; cp gpurun_out/${T}_pmc_compute.json $O/
QR_PMC_ENVS=1048576 QR_PMC_ONLY_ENV=1 bash tools/run_pmc_compute.sh ${T}_n1Mi > $O/${T}_run_pmc_compute_1Mi.log 2>&1; cp gpurun_out/${T}_n1Mi_pmc_compute.json $O/
# the measurements behind round 4's redesign: microbenchmarks, cycle probes (round-3 sources vs this build), step-kernel prologue A/B
tools/ubench/bin/valu_rate 2>&1 | grep -v amdgpu.ids > $O/${T}_valu_rate.txt
tools/ubench/bin/mfma_f16_denorm 2>&1 | grep -v amdgpu.ids > $O/${T}_mfma_f16_denorm.txt
export QR_PROBE_NOBUILD=1
(echo "## round-3 sources (f32-MFMA residual layer, rollout_stash_kernel / rollout_kernel), built with -DQR_CLOCK_PROBE"
 QR_PROBE_OLD_ABI=1 QR_PROBE_TAG=_orig python tools/clock_probe.py e2e 200 65536,131072,1048576
 QR_PROBE_OLD_ABI=1 QR_PROBE_TAG=_orig python tools/clock_probe.py e2e 20 65536
 echo "## this build (split-f16 residual layer, rollout_fast_mlp_kernel / rollout_lean_mlp_kernel)"
 python tools/clock_probe.py e2e 200 65536,131072,1048576
 python tools/clock_probe.py e2e 20 65536
 echo "## this build, general kernels (QR_ROLLOUT_FAST=0)"
 QR_ROLLOUT_FAST=0 python tools/clock_probe.py e2e 200 65536,1048576
 echo "## this build, E2E without the residual MLPs / INDI: the memory roof of the access pattern"
 QR_PROBE_NORES=1 python tools/clock_probe.py e2e 200 65536
 python tools/clock_probe.py indi 200 65536) 2>&1 | grep -v amdgpu.ids > $O/${T}_clock_probe.txt
(python tools/step_probe.py e2e; QR_PROBE_LIB=optimal_quad_control_rl_amd/_dbg/libquadrace_stepglobal.so python tools/step_probe.py e2e; python tools/step_probe.py indi) 2>&1 | grep -v amdgpu.ids > $O/${T}_step_probe.txt
ls -la $O
