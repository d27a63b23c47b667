"""Prints the fields of a bench.py JSON line that the judge reads first.  usage: python tools/show_bench.py <file>"""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "steps", "warmup", "dtype")})
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "traffic_all_pair_kernels", "kernel_ms")})
print("valu_f64", d["valu_f64"]["frac"], d["valu_f64"].get("peak_measured"))
print("utilisation", {k: v for k, v in d.get("utilisation", {}).items() if k not in ("note", "source")})
print("occupancy", {k: v for k, v in d["occupancy"].items() if k != "note"})
if "timestep" in d:
    print("timestep ms", d["timestep"]["ms_per_step"])
if "scale_ref" in d:
    print("scale_ref", {k: d["scale_ref"].get(k) for k in ("value", "ms_per_step", "pair_kernel_ms", "transport", "error")})
if "cpu_baseline" in d:
    print("cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "sample")})
if "halo" in d:
    print("halo", {k: d["halo"][k] for k in ("transport", "ranks_reported_by_transport", "peers_rank0")}, "verify", d.get("verify_rel_err"), d.get("verify_ok"))
