#!/bin/bash
# rocprofv3 kernel trace of the pair kernels for one (L, n_q): tools/rot_time.sh <tag> <lib.so> <L> <nq>  -> prints the kernel stats
set -e
tag=$1; lib=$2; L=$3; nq=$4
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -o t -- python3 "$root/tools/ab_libs.py" $lib --lmax $L --nq $nq --rounds 3 --reps 2 > /dev/null 2>&1
find "$out/${tag}_trace" -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_kernel_stats.csv" \;
rm -rf "$out/${tag}_trace"
head -4 "$out/${tag}_kernel_stats.csv" | cut -d, -f1-4
