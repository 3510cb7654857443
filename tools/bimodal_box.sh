#!/bin/bash
# tools/bimodal_box.sh: ONE line of evidence per box for the large-N INDI mode split -> gpurun_out/bimodal_box_<gpu unique id>.jsonl
#   GPU unique id | INDI 1 Mi / 64 Ki, E2E 1 Mi rollout rates and fill / copy bandwidth (tools/bimodal_probe.py) | TCC write-path counters of the
#   INDI 1 Mi rollout kernel (two rocprofv3 --pmc passes, kernel-trace only)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
uid=$(rocm-smi --showuniqueid 2>/dev/null | grep -oE "0x[0-9a-f]+" | head -1)
probe=$(timeout 300 python tools/bimodal_probe.py plain 1 2>/dev/null | tail -1)
cat > /tmp/bimodal_pmc.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch, bench
env = bench.make_env("indi", 1 << 20, 1, 0); env.reset_device()
acts = torch.rand((50, 1 << 20, 4), device="cuda") * 2 - 1
out = env.rollout_device(acts)
for _ in range(4): env.rollout_device(acts, out)
torch.cuda.synchronize()
PY
pmc() {  # $1 = tag, rest = counters
  tag=$1; shift
  rm -rf /tmp/bpmc_$tag
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/bpmc_$tag -o p -- python /tmp/bimodal_pmc.py > /tmp/bpmc_$tag.log 2>&1)
  find /tmp/bpmc_$tag -name 'p_counter_collection.csv' | head -1
}
f1=$(pmc a TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum)
f2=$(pmc b TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum)
f3=$(pmc c TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_WRITE_DRAM_sum GRBM_GUI_ACTIVE)
python - "$uid" "$probe" $f1 $f2 $f3 <<'PY' >> gpurun_out/bimodal_box_${uid}.jsonl
import csv, json, sys, collections
uid, probe, files = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {"uid": uid}
try: res["probe"] = json.loads(probe)
except Exception: res["probe_raw"] = probe[-300:]
ctr = collections.defaultdict(list)
for f in files:
    try:
        for row in csv.DictReader(open(f)):
            if "rollout" in row.get("Kernel_Name", ""):
                ctr[row["Counter_Name"]].append(float(row["Counter_Value"]))
    except Exception as e: res.setdefault("errors", []).append(repr(e))
res["pmc_per_launch"] = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in ctr.items()}   # first launch = warm-up
print(json.dumps(res))
PY
tail -1 gpurun_out/bimodal_box_${uid}.jsonl | python -c '
import json,sys
r=json.loads(sys.stdin.read()); p=r.get("probe",{})
smi=p.get("smi_after_indi",{}); print(r["uid"], {k.split("(")[1].split(")")[0] if "(" in k else k: v for k,v in smi.items() if "emperature" in k}, "INDI 1Mi %.1f again %.1f | E2E 1Mi %.1f | INDI 64Ki %.1f | fill %.2f copy %.2f" % (p["indi_1Mi"]["G_env_steps_s"], p["indi_1Mi_again"]["G_env_steps_s"], p["e2e_1Mi"]["G_env_steps_s"], p["indi_64Ki"]["G_env_steps_s"], p["fill_TBps"], p["copy_TBps_rw"]))
print({k: round(v) for k,v in r["pmc_per_launch"].items()})'
