"""Summarises rocprofv3 --pmc counter_collection CSVs for the pair kernel.
usage: python tools/pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pair_contact" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = (r["Kernel_Name"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"],
                        r["Scratch_Size"], r["Grid_Size"], r["Workgroup_Size"])
        if acc:
            print(f"# {f}")
            print("# kernel=%s vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s grid=%s wg=%s" % meta)
            for k, v in sorted(acc.items()):
                print(f"{k},{len(v)},{sum(v) / len(v):.6g}")
