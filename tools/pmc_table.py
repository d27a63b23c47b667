"""One entry of profiles/pmc_traffic.json from the summaries tools/pmc_run.sh leaves behind.
  python tools/pmc_table.py <tag>_pmc.txt <tag>_bench.json [--merge profiles/pmc_traffic.json --files profiles/<tag>_pmc.txt --commit <sha>]
Everything bench.py prints as measured-but-static utilisation comes from here: HBM-side bytes per launch of the three
pair kernels (FETCH_SIZE / WRITE_SIZE x 1024), VALU instructions per pair, VALU busy = SQ_ACTIVE_INST_VALU x 4 /
(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), LDS busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 x 256 CUs), share of LDS-active
cycles that are bank conflicts, share of the VALU instructions that are FP64 arithmetic."""
import json
import sys


def parse(path):
    out, key = {}, None
    for ln in open(path):
        ln = ln.strip()
        if ln.startswith("# kernel="):
            key = "contact" if "pair_contact" in ln else "setup" if "pair_setup" in ln else "rotate" if "pair_rotate" in ln else None
            continue
        if key and "," in ln:
            name, _, val = ln.split(",")
            out.setdefault(key, {}).setdefault(name, float(val))
    return out


def entry(pmc, bench):
    c = pmc["contact"]
    b = json.loads([ln for ln in open(bench) if ln.startswith("{")][-1])
    pairs = b["config"]["contact_pairs_rank0"]
    cfg = b["config"]
    fam = b["occupancy"]["family"]
    key = f"{cfg['particles_per_gpu']}:{cfg['lmax']}:{cfg['nq']}:{cfg['nshapes']}:{cfg['exponent']:g}:{cfg['rule']}" + (":jpoly" if fam == 1 else "")
    cu_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    e = {"fetch_bytes": int(c["FETCH_SIZE"] * 1024), "write_bytes": int(c["WRITE_SIZE"] * 1024)}
    e["traffic_bytes"] = e["fetch_bytes"] + e["write_bytes"]
    if "setup" in pmc:
        e["setup_kernel_traffic_bytes"] = int((pmc["setup"]["FETCH_SIZE"] + pmc["setup"]["WRITE_SIZE"]) * 1024)
    if "rotate" in pmc:
        e["rotate_kernel_traffic_bytes"] = int((pmc["rotate"]["FETCH_SIZE"] + pmc["rotate"]["WRITE_SIZE"]) * 1024)
        e["rotate_valu_insts_per_launch"] = pmc["rotate"]["SQ_INSTS_VALU"]
    e["traffic_all_pair_kernels_bytes"] = e["traffic_bytes"] + e.get("setup_kernel_traffic_bytes", 0) + e.get("rotate_kernel_traffic_bytes", 0)
    e["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
    e["contact_pairs"] = pairs
    e["valu_instr_per_pair"] = c["SQ_INSTS_VALU"] / pairs
    e["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cu_cycles * 1024.0)
    if "SQ_LDS_IDX_ACTIVE" in c:
        e["lds_busy"] = c["SQ_LDS_IDX_ACTIVE"] / (cu_cycles * 256.0)
        e["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "SQ_INSTS_VALU_FMA_F64" in c:
        f64 = sum(c[f"SQ_INSTS_VALU_{k}_F64"] for k in ("FMA", "MUL", "ADD", "TRANS"))
        e["fp64_instr_share"] = f64 / c["SQ_INSTS_VALU"]
        e["fp64_flop_per_pair_executed"] = (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"]
                                            + c["SQ_INSTS_VALU_TRANS_F64"]) * 64.0 / pairs
        e["int32_instr_share"] = c["SQ_INSTS_VALU_INT32"] / c["SQ_INSTS_VALU"]
    e["kernel_ms_of_the_profiled_run"] = b["roofline"]["kernel_ms"]
    # what the counters were measured ON: bench.py compares these with the library it runs and prints "stale" when they differ
    occ = b["occupancy"]
    for k in ("kernel_symbol", "kernel_hash", "ring_rows", "waves_per_pair", "vgprs", "lds_bytes_per_wave"):
        e[k] = occ.get(k)
    e["library"] = b.get("library")
    return key, e


if __name__ == "__main__":
    if sys.argv[1] == "--merge-json":   # pmc_table.py --merge-json profiles/pmc_traffic.json <tag>_table.json ...: entries made on the box
        tab = json.load(open(sys.argv[2]))
        for f in sys.argv[3:]:
            for k, e in json.load(open(f)).items():
                tab[k] = e
                print(f"merged {k} from {f}")
        json.dump(tab, open(sys.argv[2], "w"), indent=1)
        sys.exit(0)
    pmc = parse(sys.argv[1])
    key, e = entry(pmc, sys.argv[2])
    if "--files" in sys.argv:
        e["files"] = sys.argv[sys.argv.index("--files") + 1]
    if "--commit" in sys.argv:   # the commit whose build was profiled (git rev-parse --short HEAD; "+dirty" if the tree had changes)
        e["commit"] = sys.argv[sys.argv.index("--commit") + 1]
    if "--merge" in sys.argv:
        path = sys.argv[sys.argv.index("--merge") + 1]
        tab = json.load(open(path))
        tab[key] = e
        json.dump(tab, open(path, "w"), indent=1)
        print(f"merged {key} into {path}")
    else:
        print(json.dumps({key: e}, indent=1))
