#!/bin/bash
# kernel-level timing of the PPO update (rocprofv3 --kernel-trace --stats), default kernel and QR_PPO_GRAD4=1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in 8 4; do
  rm -rf /tmp/prof_ppo$v
  if [ $v = 4 ]; then export QR_PPO_GRAD4=1; else unset QR_PPO_GRAD4; fi
  rocprofv3 --kernel-trace --stats -d /tmp/prof_ppo$v -o ppo -- python $R/tools/bench_ppo_update.py --iters 200 > /dev/null 2>&1
  f=$(find /tmp/prof_ppo$v -name "*kernel_stats.csv" | head -1)
  echo "== grad$v: $f"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "ppo_" in n:
        print(f'{int(r["Calls"]):6d} avg {float(r["AverageNs"]):9.0f} ns  min {float(r["MinNs"]):8.0f}  max {float(r["MaxNs"]):8.0f}  {n[:70]}')
PY
done
