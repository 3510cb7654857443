#!/bin/bash
# GPU box, round 6: the reference's own recipe (tools/reference_recipe_run.py, 1.03e9 env-steps per run)
#   (1) seed 0 on the default matrix-core path: must reproduce profiles/r05b_refrecipe_seed0.json digit for digit (arithmetic-neutral round)
#   (2) seeds 0..9 with the reference-precision forward in the collect phase (--precision f32-collect): do the seeds that missed the
#       reference's lap band (2 and 7) still miss when the collected actions / log-probabilities are float32-class?
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/refrecipe_r06; mkdir -p $O
python tools/reference_recipe_run.py --seed 0 --out $O/r06_refrecipe_seed0.json > $O/seed0.log 2>&1
for s in ${QR_SEEDS:-0 1 2 3 4 5 6 7 8 9}; do
  python tools/reference_recipe_run.py --seed $s --precision f32-collect --out $O/r06_f32collect_seed$s.json > $O/f32collect_seed$s.log 2>&1
done
python - <<'PY'
import json, glob, os
O = "gpurun_out/refrecipe_r06"
def line(f):
    d = json.load(open(f)); fin = d["final"] or {}; best = d["best_checkpoint"] or {}
    return "%-28s seed %d %-13s %6.1f s  %.2f M steps/s  final lap %s crashes %.3f | best lap %s | reaches <= 2.6 s after %s s" % (
        os.path.basename(f), d["seed"], d["precision"], d["train_seconds"], d["env_steps_per_s"] / 1e6, fin.get("flying_lap"), fin.get("crashes_per_12s", -1),
        best.get("flying_lap"), d["reaches_reference_level_after_s"])
for f in sorted(glob.glob(O + "/r06_*.json")): print(line(f))
old = "profiles/r05b_refrecipe_seed0.json"
if os.path.exists(old) and os.path.exists(O + "/r06_refrecipe_seed0.json"):
    a, b = json.load(open(old)), json.load(open(O + "/r06_refrecipe_seed0.json"))
    same = all(x[k] == y[k] for x, y in zip(a["curve"], b["curve"]) for k in ("flying_lap", "gates_per_12s", "crashes_per_12s", "env_steps"))
    print("seed 0 vs round 5 (%s): evaluation curve %s (%d points)" % (old, "IDENTICAL digit for digit" if same and len(a["curve"]) == len(b["curve"]) else "DIFFERS", len(b["curve"])))
PY
