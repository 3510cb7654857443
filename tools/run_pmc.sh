#!/bin/bash
# GPU box: collect the FETCH_SIZE / WRITE_SIZE passes for both variants and summarise them (profiles/pmc_summary.json)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
for v in e2e indi; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${v}_fetch -o f -- python tools/pmc_probe.py $v > gpurun_out/pmc_${v}_f.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${v}_write -o w -- python tools/pmc_probe.py $v > gpurun_out/pmc_${v}_w.log 2>&1
  python tools/pmc_traffic.py $v 65536 gpurun_out/pmc_${v}_fetch/f_counter_collection.csv gpurun_out/pmc_${v}_write/w_counter_collection.csv gpurun_out/profiles/pmc_summary.json
done
