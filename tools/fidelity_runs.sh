#!/bin/bash
# GPU box: the config-5 fidelity runs behind DESIGN.md section 8 "Which recipe trains?" (results: profiles/r03_ppo_e2e_*.json).
# E2E model (residual MLPs + disturbances), 4-gate square track, reference hyper-parameters (R:783-795: 3 x 120 ReLU nets, gamma 0.999,
# 10 epochs, lr 3e-4, target_kl None), deterministic evaluation of 4 096 fresh envs (tools/train_ppo.py).
#   bash tools/fidelity_runs.sh            # everything (~45 min of GPU time)
#   source tools/fidelity_runs.sh lib; run_geo 4096 512 0 --eval-final --lr-final 1.0     # one run
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/fid
summ() { python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1 seed', j['seed'], j['evaluated'][:5], 'train_s %.1f' % j['train_seconds'], 'flying lap %.3f' % j['eval_flying_lap_seconds'], 'first %.2f' % j['eval_lap_seconds']['lap1'], 'gates12 %.2f crashes %.3f' % (j['eval_gates_per_12s'], j['eval_crashes_per_12s']), 'updates', j['updates'], 'skipped', j['skipped_nonfinite'])"; }
# run_geo ENVS NSTEPS SEED [train_ppo.py flags...]  -> gpurun_out/fid/r03_ppo_e2e_<tag>_seed<SEED>.json (TAG env var names the series)
run_geo() {
  envs=$1; ns=$2; seed=$3; shift 3
  tag=${TAG:-constlr_final_${envs}x${ns}}
  python tools/train_ppo.py --variant e2e --track square --envs $envs --steps ${STEPS:-3e9} --n-steps $ns --epochs 10 --minibatches ${MB:-128} --lr 3e-4 \
     --target-kl 1e9 --gamma 0.999 --fused --native-update --seed $seed "$@" --out gpurun_out/fid/r03_ppo_e2e_${tag}_seed$seed.json 2>&1 | tail -1 | summ $tag
}
[ "$1" = "lib" ] && return 0
CF="--lr-final 1.0 --eval-final"                                       # the reference's optimiser settings, FINAL policy
for s in 0 1 2 3 4; do run_geo 4096 512 $s $CF; done
for s in 0 1 2; do run_geo 2048 1024 $s $CF; done
for s in 0 1; do run_geo 16384 128 $s $CF; done
for s in 0 1 2 3 4; do TAG=constlr_final_mb16384_3e9 run_geo 65536 32 $s $CF; done
for s in 0 1 2 3 4; do TAG=constlr_final_mb65536_3e9 MB=32 run_geo 65536 32 $s $CF; done
for s in 0 1; do TAG=constlr_final_mb65536_1e10 MB=32 STEPS=1e10 run_geo 65536 32 $s $CF; done
for s in 0 1; do TAG=const_best run_geo 65536 32 $s --lr-final 1.0; done                        # best checkpoint by training statistic
for s in 0 1 2; do TAG=decay_final run_geo 65536 32 $s --lr-final 0.1 --eval-final; done       # lr 3e-4 -> 3e-5, final policy
for s in 0 1; do TAG=decay_best run_geo 65536 32 $s --lr-final 0.1; done                        # round 2's protocol
