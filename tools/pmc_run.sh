#!/bin/bash
# rocprofv3 passes of one bench workload on the GPU box: kernel trace + stats, then PMC counters in SEPARATE passes
# (MI355X_MICROARCH.md: TCC slots do not hold FETCH_SIZE and WRITE_SIZE together; never combine --pmc with a trace).
#   tools/pmc_run.sh <tag> [bench.py arguments...]      -> gpurun_out/<tag>_*.{csv,txt,json}
# The program after `--` is python3 itself (no env / bash hop: the profiler's preload has initialised the GPU).
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--cpu-seconds 0 --ts-steps 0 --peak-ms 0 --scale-ref 0 --configs 0 --host-path 0 --steps 10 $*"
python3 "$root/bench.py" $args > "$out/${tag}_bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -o t -- python3 "$root/bench.py" $args > /dev/null
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $set --output-format csv -d "$out/${tag}_pmc_$name" -o p -- python3 "$root/bench.py" $args > /dev/null
done
python3 "$root/tools/pmc_summary.py" "$out"/${tag}_pmc_* > "$out/${tag}_pmc.txt"
# the table entry bench.py reports from (profiles/pmc_traffic.json): printed, to be pasted / merged with tools/pmc_table.py --merge
# .commit: `git rev-parse --short HEAD` written before the gpurun call (the box has no .git)
commit=$(cat "$root/.commit" 2>/dev/null || echo unknown)
python3 "$root/tools/pmc_table.py" "$out/${tag}_pmc.txt" "$out/${tag}_bench.json" --files "profiles/${tag}_pmc.txt" --commit "$commit" > "$out/${tag}_table.json" || true
find "$out/${tag}_trace" -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_kernel_stats.csv" \;
# the raw per-dispatch CSVs are large: keep the summaries
rm -rf "$out"/${tag}_pmc_* "$out/${tag}_trace"
echo "== $tag"; cat "$out/${tag}_pmc.txt"; head -5 "$out/${tag}_kernel_stats.csv"
