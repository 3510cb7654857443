#!/bin/bash
# GPU box: collect the FETCH_SIZE / WRITE_SIZE passes for both variants (and, for E2E, at 1 Mi envs too) and summarise them by kernel
# symbol -> gpurun_out/profiles/<tag>_pmc_traffic.json   (copy to profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
OUT=gpurun_out/profiles/${QR_TAG:-r06}_pmc_traffic.json; rm -f $OUT
for cfg in "e2e 65536" "indi 65536" "e2e 1048576"; do
  set -- $cfg; v=$1; n=$2
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${v}_${n}_fetch -o f -- python tools/pmc_probe.py $v $n > gpurun_out/pmc_${v}_${n}_f.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${v}_${n}_write -o w -- python tools/pmc_probe.py $v $n > gpurun_out/pmc_${v}_${n}_w.log 2>&1
  QR_COMMIT=$QR_COMMIT python tools/pmc_traffic.py $v $n $(find gpurun_out/pmc_${v}_${n}_fetch -name 'f_counter_collection.csv') $(find gpurun_out/pmc_${v}_${n}_write -name 'w_counter_collection.csv') $OUT
done
