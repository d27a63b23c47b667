"""Summarises rocprofv3 --pmc counter_collection CSVs for the pair kernels (contact kernel, per-pair set-up kernel, rotation kernel).
usage: python tools/pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in sorted(glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            for key in ("pair_contact", "pair_setup", "pair_rotate"):
                if key in r["Kernel_Name"]:
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta[key] = (r["Kernel_Name"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"],
                                 r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"])
        for key in acc:
            # the register / LDS columns of this header are rocprofv3's, NOT the code object's: see profiles/README.md
            print("# kernel=%s vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s grid=%s wg=%s" % meta[key])
            for k, v in sorted(acc[key].items()):
                print(f"{k},{len(v)},{sum(v) / len(v):.6g}")
