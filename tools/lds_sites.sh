#!/bin/bash
# Attribution of LDS-pipe cycles and bank conflicts to the phases of pair_contact_kernel: one rocprofv3 --pmc pass per
# timing-only ablation build (make -C lammps-spherharm_amd/csrc abl A=1|4|2|3: stop after the prologue / after
# particle j's table build / after the ring tables / phase 1 only) and one of the shipped library; differences between
# consecutive builds are the sites' shares.
#   tools/lds_sites.sh <tag> [ab_libs.py arguments, e.g. --lmax 6 --nq 16]     -> gpurun_out/<tag>_lds_sites.txt
# The program after `--` is python3 itself (no env / bash hop: the profiler's preload has initialised the GPU).
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
: > "$out/${tag}_lds_sites.txt"
for lib in libshpair_abl1.so libshpair_abl4.so libshpair_abl2.so libshpair_abl3.so libshpair.so; do
  [ -f "$root/lammps-spherharm_amd/shpair/$lib" ] || continue
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d "$out/${tag}_ls_$lib" -o p -- python3 "$root/tools/ab_libs.py" $lib --rounds 10 "$@" > /dev/null
  echo "## $lib" >> "$out/${tag}_lds_sites.txt"
  python3 "$root/tools/pmc_summary.py" "$out/${tag}_ls_$lib" | grep -A8 "pair_contact" >> "$out/${tag}_lds_sites.txt"
  rm -rf "$out/${tag}_ls_$lib"
done
cat "$out/${tag}_lds_sites.txt"
