"""Static instruction profile of one kernel of a hipcc -S listing, per basic block (no GPU needed).
  hipcc <flags of csrc/Makefile> -DSHP_L=6 --cuda-device-only -S pair_kernels_inst.hip -o /tmp/L6.s
  python tools/isa_blocks.py /tmp/L6.s Li6ELb1ELb0E [--min 8] [--dump LABEL]
Columns: block label, first line, VALU total, of which FP64 arithmetic (fma/mul/add/rcp/rsq/...), moves (v_mov /
v_accvgpr), selects + compares (v_cndmask, v_cmp), lane ops (readlane/writelane/readfirstlane/permute), other VALU;
then DS, VMEM (global/flat/buffer), SMEM, SALU.  Multiply a block by how often it runs (tools/kernel_stats.py) to see
where SQ_INSTS_VALU comes from."""
import re
import sys
from collections import OrderedDict


def classify(op):
    if op.startswith("v_"):
        if re.match(r"v_(fma|mul|add|fmac|rcp|rsq|sqrt|max|min|ldexp|frexp|trig|div|rndne|floor|fract|cvt).*f64", op) or op in (
                "v_cvt_f64_i32", "v_cvt_f64_u32", "v_cvt_i32_f64", "v_cvt_u32_f64"):
            return "f64"
        if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_swap"):
            return "mov"
        if op.startswith("v_cndmask") or op.startswith("v_cmp"):
            return "sel"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_bpermute", "v_mbcnt")):
            return "lane"
        return "valu"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    minv = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith("_Z") and key in ln and ln.rstrip().split(":")[0].endswith("E"))
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = dict(line=start + 1, note="")
    for i in range(start + 1, len(lines)):
        ln = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = dict(line=i + 1, note=m.group(2).strip(" ;"))
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")):
            if s.startswith(".Lfunc_end"):
                break
            continue
        op = s.split()[0]
        c = classify(op)
        b = blocks[cur]
        b[c] = b.get(c, 0) + 1
        if dump == cur:
            print(f"{i + 1:6d} {s}")
    if dump:
        return
    hdr = f"{'block':12s} {'line':>6s} {'VALU':>5s} {'f64':>5s} {'mov':>4s} {'sel':>4s} {'lane':>4s} {'oth':>4s} | {'ds':>4s} {'vmem':>4s} {'smem':>4s} {'salu':>4s}  note"
    print(hdr)
    tot = {}
    for name, b in blocks.items():
        valu = sum(b.get(k, 0) for k in ("f64", "mov", "sel", "lane", "valu"))
        for k in ("f64", "mov", "sel", "lane", "valu", "ds", "vmem", "smem", "salu"):
            tot[k] = tot.get(k, 0) + b.get(k, 0)
        if valu < minv:
            continue
        print(f"{name:12s} {b['line']:6d} {valu:5d} {b.get('f64', 0):5d} {b.get('mov', 0):4d} {b.get('sel', 0):4d} {b.get('lane', 0):4d} "
              f"{b.get('valu', 0):4d} | {b.get('ds', 0):4d} {b.get('vmem', 0):4d} {b.get('smem', 0):4d} {b.get('salu', 0):4d}  {b['note'][:60]}")
    valu = sum(tot.get(k, 0) for k in ("f64", "mov", "sel", "lane", "valu"))
    print(f"{'TOTAL':12s} {'':6s} {valu:5d} {tot.get('f64', 0):5d} {tot.get('mov', 0):4d} {tot.get('sel', 0):4d} {tot.get('lane', 0):4d} "
          f"{tot.get('valu', 0):4d} | {tot.get('ds', 0):4d} {tot.get('vmem', 0):4d} {tot.get('smem', 0):4d} {tot.get('salu', 0):4d}")


if __name__ == "__main__":
    main()
