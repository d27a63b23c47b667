#!/bin/bash
# Diagnostic PMC passes of one bench workload (one rocprofv3 run per counter set, no trace):
#   tools/pmc_diag.sh <tag> "<set 1>;<set 2>;..." [bench.py arguments...]   -> gpurun_out/<tag>_diag.txt
# The program after `--` is python3 itself (no env / bash hop: the profiler's preload has initialised the GPU).
set -e
tag=$1; sets=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--cpu-seconds 0 --ts-steps 0 --peak-ms 0 --scale-ref 0 --configs 0 --host-path 0 --steps 10 $*"
i=0
IFS=';' read -ra SETS <<< "$sets"
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$out/${tag}_diag_$i" -o p -- python3 "$root/bench.py" $args > /dev/null
done
python3 "$root/tools/pmc_summary.py" "$out"/${tag}_diag_* > "$out/${tag}_diag.txt"
rm -rf "$out"/${tag}_diag_[0-9]*
cat "$out/${tag}_diag.txt"
