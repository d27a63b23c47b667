#!/bin/bash
# Interleaved A/B of one wave per pair ("split" 0) against two ("split" 1) over the orders the two-wave kernels are
# compiled for:  tools/split_matrix.sh > gpurun_out/<tag>_split_matrix.txt
root=$(cd "$(dirname "$0")/.." && pwd)
for L in 7 8 9 10 11 12; do
  for nq in 8 12 16 20 24 32; do
    echo "== L $L nq $nq"
    python3 "$root/tools/ab_libs.py" libshpair.so libshpair.so --split 0 1 --jpoly 1 --lmax $L --nq $nq --rounds 4 2>&1 | grep -v amdgpu.ids
  done
done
