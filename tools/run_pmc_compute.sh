#!/bin/bash
# GPU box: the four SQ counter passes of tools/pmc_compute.py (compute-side roofline evidence) -> gpurun_out/pmc_compute.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02}
A=$(python tools/pmc_compute.py groups | sed -n 1p)
B=$(python tools/pmc_compute.py groups | sed -n 2p)
timeout 900 rocprofv3 --kernel-trace --pmc $A --output-format csv -d gpurun_out/pmcc_a -o p -- python tools/pmc_compute.py probe > gpurun_out/pmcc_a.log 2>&1
C=$(python tools/pmc_compute.py groups | sed -n 3p)
timeout 900 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmcc_c -o p -- python tools/pmc_compute.py probe > gpurun_out/pmcc_c.log 2>&1
D=$(python tools/pmc_compute.py groups | sed -n 4p)
timeout 900 rocprofv3 --kernel-trace --pmc $D --output-format csv -d gpurun_out/pmcc_d -o p -- python tools/pmc_compute.py probe > gpurun_out/pmcc_d.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc $B --output-format csv -d gpurun_out/pmcc_b -o p -- python tools/pmc_compute.py probe > gpurun_out/pmcc_b.log 2>&1
python tools/pmc_compute.py summarise $(find gpurun_out/pmcc_a gpurun_out/pmcc_b gpurun_out/pmcc_c gpurun_out/pmcc_d -name '*counter_collection.csv') gpurun_out/${TAG}_pmc_compute.json
