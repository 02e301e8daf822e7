#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter per pass) into per-kernel, per-launch averages.

usage: pmc_summary.py OUT.json "workload tag" DIR [DIR ...]      (each DIR holds *_counter_collection.csv of one pass)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out, tag, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                per_dispatch = defaultdict(float)
                names = {}
                for row in csv.DictReader(fh):
                    key = (row["Dispatch_Id"], row["Counter_Name"])
                    per_dispatch[key] += float(row["Counter_Value"])   # one row per XCD/instance: sum them
                    names[row["Dispatch_Id"]] = row["Kernel_Name"]
                for (disp, cname), v in per_dispatch.items():
                    k = names[disp].split("(")[0].replace("void ", "")
                    a = acc[k][cname]
                    a[0] += v
                    a[1] += 1
    kernels = {}
    for k, cs in sorted(acc.items()):
        e = {}
        for cname, (tot, n) in cs.items():
            e[f"{cname}_KB_per_launch"] = round(tot / max(n, 1), 2)
            e["launches"] = n
        kernels[k] = e
    json.dump({"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py --steps 1 --warmup 1 "
                          "--no-cpu-baseline --no-extra (two separate passes; the kernel's launches include the output-less replay launches of the roofline measurement)",
               "note": "RAW counter values in KB (rocprofv3 FETCH_SIZE / WRITE_SIZE), two separate passes.  MI355X_MICROARCH.md (HBM): on gfx950 "
                       "FETCH_SIZE reports exactly half of the bytes of a 16-B-per-lane streaming read -- consumers double it for kernels "
                       "whose fetches are such streams (bench.py does for the forward kernel's weight loads); WRITE_SIZE is exact for "
                       "16-B-per-lane stores; other access widths are uncalibrated",
               "workload": tag, "kernels": kernels}, open(out, "w"), indent=1)
    for k, e in kernels.items():
        if "eaz" in k:
            print(k, e)


if __name__ == "__main__":
    main()
