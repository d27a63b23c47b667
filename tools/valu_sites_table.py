"""Per-phase VALU instruction counts from the raw summaries tools/valu_sites.sh collects (one block per ablation build).
  python tools/valu_sites_table.py gpurun_out/<tag>_valu_sites.txt [pairs]
Prints, per phase (difference of consecutive builds) and per contact pair: VALU instructions, FP64 arithmetic (FMA +
MUL + ADD + TRANS), INT32, the rest (moves, compares, selects, conversions), and the non-FP64 share."""
import re
import sys

PHASES = [("libshpair_abl1.so", "prologue"), ("libshpair_abl4.so", "particle j's table"), ("libshpair_abl2.so", "ring tables"),
          ("libshpair_abl3.so", "phase 1"), ("libshpair.so", "phase 2 + epilogue")]


def parse(path):
    out, cur = {}, None
    for ln in open(path):
        ln = ln.strip()
        m = re.match(r"## (\S+)", ln)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        if cur and re.match(r"^[A-Z0-9_]+,\d+,", ln):
            name, n, val = ln.split(",")
            out[cur][name] = float(val)
            if name == "SQ_INSTS_VALU":
                out[cur]["_launches"] = int(n)
        m = re.search(r"grid=(\d+)", ln)
        if cur and m:
            out[cur]["_grid"] = int(m.group(1))
    return out


def main():
    tab = parse(sys.argv[1])
    pairs = float(sys.argv[2]) if len(sys.argv) > 2 else None
    if pairs is None:   # one wave (64 work-items) per slot, or 128 with two waves per pair: the caller may pass the contact-pair count
        g = next(iter(tab.values())).get("_grid", 0)
        pairs = g / 64.0
    prev = {}
    print(f"# per list slot ({pairs:.0f} slots; SQ_INSTS_* are wave instructions)")
    print(f"{'phase':22s} {'VALU':>8s} {'FP64':>8s} {'INT32':>8s} {'rest':>8s} {'non-FP64':>9s} {'SALU':>8s}")
    tot = None
    for lib, name in PHASES:
        if lib not in tab:
            continue
        c = tab[lib]
        f64 = sum(c.get(f"SQ_INSTS_VALU_{k}_F64", 0.0) for k in ("FMA", "MUL", "ADD", "TRANS"))
        cur = {"valu": c["SQ_INSTS_VALU"], "f64": f64, "int": c.get("SQ_INSTS_VALU_INT32", 0.0), "salu": c.get("SQ_INSTS_SALU", 0.0)}
        d = {k: (cur[k] - prev.get(k, 0.0)) / pairs for k in cur}
        rest = d["valu"] - d["f64"] - d["int"]
        print(f"{name:22s} {d['valu']:8.1f} {d['f64']:8.1f} {d['int']:8.1f} {rest:8.1f} {100 * (1 - d['f64'] / max(d['valu'], 1e-9)):8.1f}% {d['salu']:8.1f}")
        prev, tot = cur, cur
    if tot:
        rest = tot["valu"] - tot["f64"] - tot["int"]
        print(f"{'all':22s} {tot['valu'] / pairs:8.1f} {tot['f64'] / pairs:8.1f} {tot['int'] / pairs:8.1f} {rest / pairs:8.1f} "
              f"{100 * (1 - tot['f64'] / tot['valu']):8.1f}% {tot['salu'] / pairs:8.1f}")


if __name__ == "__main__":
    main()
