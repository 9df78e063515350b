#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: sum of each counter, launches,
total duration; derived MFMA utilisation / effective clock for the MI355X (256 CUs, 4 SIMDs/CU).
    python tools/pmc_summary.py <dir-with-csv> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    dur = defaultdict(float)
    seen = defaultdict(set)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if want not in k:
                    continue
                k = k.split("(")[0][-70:]
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                did = (f, row["Dispatch_Id"])
                if did not in seen[k]:
                    seen[k].add(did)
                    if "Start_Timestamp" in row and row["Start_Timestamp"]:
                        dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    for k in agg:
        n = len(seen[k])
        print(f"== {k}: {n} dispatches, {dur[k] / 1e6:.3f} ms total")
        c = agg[k]
        for name in sorted(c):
            print(f"   {name:34s} {c[name]:.6g}")
        if "GRBM_GUI_ACTIVE" in c and dur[k] > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md 'DVFS give-back')
            print(f"   -> effective clock ~ {c['GRBM_GUI_ACTIVE'] / 8 / (dur[k] * 1e-9) / 1e6:.0f} MHz")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
                print(f"   -> MFMA busy / (active cycles x 1024 SIMDs) = {100 * util:.1f} %")
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c and dur[k] > 0:
            print(f"   -> fp64 MFMA flop rate = {c['SQ_INSTS_VALU_MFMA_MOPS_F64'] * 512 / (dur[k] * 1e-9) / 1e12:.2f} TFlop/s")
        if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
            print(f"   -> LDS bank conflict cycles / LDS active = {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1f} %")
        if "FETCH_SIZE" in c:
            print(f"   -> FETCH_SIZE x2 (gfx950 correction) = {c['FETCH_SIZE'] * 2 * 1024 / 1e9:.3f} GB")
        if "WRITE_SIZE" in c:
            print(f"   -> WRITE_SIZE = {c['WRITE_SIZE'] * 1024 / 1e9:.3f} GB")


if __name__ == "__main__":
    main()
