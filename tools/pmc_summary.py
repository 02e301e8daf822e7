#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter per pass) into per-kernel, per-launch averages.

usage: pmc_summary.py OUT.json "workload tag" DIR [DIR ...]      (each DIR holds *_counter_collection.csv of one pass)
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTE_COUNTERS = ("FETCH_SIZE", "WRITE_SIZE")  # rocprofv3 reports these in KB; every other counter is a plain count (cycles, instructions, percent)


def source_hash():
    """= e-alphazero_amd/_lib.py source_hash() = the library's eaz_build_hash(): the code these counters were recorded on"""
    csrc = os.path.join(ROOT, "e-alphazero_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")), key=os.path.basename)
    files.append(os.path.join(ROOT, "include", "eaz.h"))
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


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
            e[f"{cname}_KB_per_launch" if cname in BYTE_COUNTERS else f"{cname}_per_launch"] = round(tot / max(n, 1), 2)
            e["launches"] = n
        kernels[k] = e
    counters = sorted({c for cs in acc.values() for c in cs})
    traffic = all(c in BYTE_COUNTERS for c in counters)
    note = ("RAW counter values in KB (rocprofv3 FETCH_SIZE / WRITE_SIZE), two separate passes.  MI355X_MICROARCH.md (HBM): on gfx950 "
            "FETCH_SIZE reports exactly half of the bytes of a 16-B-per-lane streaming read -- consumers double it for kernels "
            "whose fetches are such streams (bench.py does for the forward kernel's weight loads); WRITE_SIZE is exact for "
            "16-B-per-lane stores; other access widths are uncalibrated") if traffic else \
           ("RAW counter values per launch, summed over the XCDs / SEs rocprofv3 reports separately, one --pmc pass per counter: cycle and "
            "instruction counts are plain numbers (NOT bytes), MfmaUtil is rocprofv3's derived percentage averaged over its instances")
    json.dump({"command": "rocprofv3 --kernel-trace --pmc <%s> --output-format csv -- python3 bench.py [--workload ...] --steps 1 --warmup 1 "
                          "--no-cpu-baseline --no-extra (one pass per counter; the kernel's launches include the output-less replay launches of the "
                          "roofline measurement)" % "|".join(counters),
               "note": note, "source_hash": source_hash(),
               "source_hash_note": "SHA-256 prefix of csrc/*.hip, csrc/*.hpp, include/eaz.h at recording time (= eaz_build_hash() of the library that ran)",
               "workload": tag, "kernels": kernels}, open(out, "w"), indent=1)
    for k, e in kernels.items():
        if "eaz" in k:
            print(k, e)


if __name__ == "__main__":
    main()
