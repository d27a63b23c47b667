#!/bin/bash
# The five workloads bench.py's static PMC table covers, each through tools/pmc_run.sh (on the GPU box):
#   tools/pmc_refresh.sh <prefix>     -> gpurun_out/<prefix>_{headline,mixed4,L12,L4,weighted}_{bench.json,pmc.txt,kernel_stats.csv,table.json}
# Afterwards, here:  python tools/pmc_table.py --merge-json profiles/pmc_traffic.json gpurun_out/<prefix>_*_table.json
#                    cp gpurun_out/<prefix>_*_{bench.json,pmc.txt,kernel_stats.csv} profiles/
set -e
p=$1
root=$(cd "$(dirname "$0")/.." && pwd)
bash "$root/tools/pmc_run.sh" ${p}_headline > /dev/null 2>&1
bash "$root/tools/pmc_run.sh" ${p}_mixed4 --nshapes 4 > /dev/null 2>&1
bash "$root/tools/pmc_run.sh" ${p}_L12 --lmax 12 --nq 32 > /dev/null 2>&1
bash "$root/tools/pmc_run.sh" ${p}_L4 --lmax 4 --nq 10 > /dev/null 2>&1
bash "$root/tools/pmc_run.sh" ${p}_weighted --rule weighted > /dev/null 2>&1
for w in headline mixed4 L12 L4 weighted; do echo "== $w"; head -4 "$root/gpurun_out/${p}_${w}_kernel_stats.csv"; done
