#!/bin/bash
# config-5 fidelity: reference optimiser settings (constant lr 3e-4, target_kl None, 10 epochs, gamma 0.999), FINAL policy evaluated
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/fid
MB=${MB:-128}; STEPS=${STEPS:-3e9}; TAG=${TAG:-mb16384}
for seed in ${SEEDS:-0 1 2 3 4}; do
  python tools/train_ppo.py --variant e2e --track square --envs 65536 --steps $STEPS --n-steps 32 --epochs 10 --minibatches $MB --lr ${LR:-3e-4} \
     --lr-final 1.0 --target-kl 1e9 --gamma 0.999 --fused --native-update --eval-final --seed $seed \
     --out gpurun_out/fid/r03_ppo_e2e_constlr_final_${TAG}_${STEPS}_seed$seed.json > gpurun_out/fid/log_${TAG}_$seed.txt 2>&1
  tail -1 gpurun_out/fid/log_${TAG}_$seed.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$TAG seed', j['seed'], 'train_s %.1f' % j['train_seconds'], 'Msteps/s %.1f' % j['train_Msteps_per_s'], 'flying lap %.3f' % j['eval_flying_lap_seconds'], 'first', '%.2f' % j['eval_lap_seconds']['lap1'], 'gates12 %.2f crashes %.3f' % (j['eval_gates_per_12s'], j['eval_crashes_per_12s']))"
done
