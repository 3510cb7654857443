#!/bin/bash
# GPU box, round 6: the reference's recipe with the reference's precision END TO END (--precision f32: f32-class policy forward in the fused collect
# kernel + f32-class gradient kernels qr_ppo_grad_f32class + the f32 Adam kernel), seeds QR_SEEDS.  Each run ends at the first evaluation (every
# 4e7 env-steps) that reaches the reference's level, or after the full 1.03e9 steps: "does this seed reach it" is the question.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/refrecipe_f32_r06; mkdir -p $O
for s in ${QR_SEEDS:-0 1 2 3 4 5 6 7 8 9}; do
  python tools/reference_recipe_run.py --seed $s --precision f32 --stop-when-reached --out $O/r06_f32_seed$s.json > $O/f32_seed$s.log 2>&1
done
python - <<'PY'
import json, glob, os
O = "gpurun_out/refrecipe_f32_r06"
for f in sorted(glob.glob(O + "/r06_f32_seed*.json")):
    d = json.load(open(f)); fin = d["final"] or {}; best = d["best_checkpoint"] or {}
    print("%-22s seed %d %-4s %7.1f s  %.2f M steps/s  %4d M steps %s | last evaluation lap %s crashes %.3f | best lap %s | reaches <= 2.6 s after %s s" % (
        os.path.basename(f), d["seed"], d["precision"], d["train_seconds"], d["env_steps_per_s"] / 1e6, d["train_steps"] // 10 ** 6,
        "(stopped when reached)" if d["stopped_when_reached"] else "(full run)", fin.get("flying_lap"), fin.get("crashes_per_12s", -1),
        best.get("flying_lap"), d["reaches_reference_level_after_s"]))
PY
