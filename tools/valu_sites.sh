#!/bin/bash
# Attribution of VALU issue slots to the phases of pair_contact_kernel: one rocprofv3 --pmc pass per timing-only
# ablation build (make -C lammps-spherharm_amd/csrc abl A=1|4|2|3: stop after the prologue / after particle j's table /
# after the ring tables / phase 1 only) and one of the shipped library; differences between consecutive builds are the
# phases' instruction counts, split into FP64 arithmetic, 32-bit integer and the rest (moves, compares, selects).
#   tools/valu_sites.sh <tag> [ab_libs.py arguments, e.g. --lmax 6 --nq 16]     -> gpurun_out/<tag>_valu_sites.txt
# The program after `--` is python3 itself (no env / bash hop: the profiler's preload has initialised the GPU).
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
: > "$out/${tag}_valu_sites.txt"
for lib in libshpair_abl1.so libshpair_abl4.so libshpair_abl2.so libshpair_abl3.so libshpair.so; do
  [ -f "$root/lammps-spherharm_amd/shpair/$lib" ] || continue
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU \
    --output-format csv -d "$out/${tag}_vs_$lib" -o p -- python3 "$root/tools/ab_libs.py" $lib --rounds 4 --reps 1 "$@" > /dev/null
  echo "## $lib" >> "$out/${tag}_valu_sites.txt"
  python3 "$root/tools/pmc_summary.py" "$out/${tag}_vs_$lib" | grep -A8 "pair_contact" >> "$out/${tag}_valu_sites.txt"
  rm -rf "$out/${tag}_vs_$lib"
done
python3 "$root/tools/valu_sites_table.py" "$out/${tag}_valu_sites.txt" | tee -a "$out/${tag}_valu_sites.txt"
