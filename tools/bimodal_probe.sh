#!/bin/bash
# tools/bimodal_probe.sh [plain-runs] [slab-runs]: N fresh processes of tools/bimodal_probe.py on THIS box -> gpurun_out/bimodal_probe.jsonl
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/bimodal_probe.jsonl
echo "# box $(hostname) $(date -u +%FT%TZ) $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' | tr -s ' ')" >> $out
rocm-smi --showmemuse --showclocks 2>/dev/null | grep -iE "sclk|mclk|fclk|GPU\[0\]" | head -8 | sed 's/^/# /' >> $out
for i in $(seq 1 ${1:-6}); do timeout 300 python tools/bimodal_probe.py plain $i 2>/dev/null | tail -1 >> $out; done
for i in $(seq 1 ${2:-3}); do timeout 300 python tools/bimodal_probe.py slab $i 2>/dev/null | tail -1 >> $out; done
python - <<'PY'
import json
for line in open("gpurun_out/bimodal_probe.jsonl"):
    if line.startswith("#") or not line.strip(): print(line.rstrip()); continue
    r = json.loads(line)
    smi = r.get("smi_after_indi", {})
    clk = " ".join(f"{k.split('(')[0].strip()}={v}" for k, v in smi.items() if "sclk" in k.lower() or "power" in k.lower() and "socket" in k.lower())
    print(f"{r['mode']:5s} #{r['tag']}: fill {r['fill_TBps']:.2f} TB/s copy {r['copy_TBps_rw']:.2f} TB/s | INDI 1Mi {r['indi_1Mi']['G_env_steps_s']:.1f} ({r['indi_1Mi']['lo']:.1f}-{r['indi_1Mi']['hi']:.1f}) again {r['indi_1Mi_again']['G_env_steps_s']:.1f} | E2E 1Mi {r['e2e_1Mi']['G_env_steps_s']:.1f} | INDI 64Ki {r['indi_64Ki']['G_env_steps_s']:.1f} | {clk} | obs ptr GiB {r['indi_1Mi']['ptr_GiB'][1]}")
PY
