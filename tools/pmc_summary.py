#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one counter per pass) per kernel family.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -o p -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -o p -- python3 bench.py ...
    python tools/pmc_summary.py out/fetch out/write > profiles/rNN_pmc_....json

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Corrections (MI355X_MICROARCH.md, HBM section) are NOT
applied here: on gfx950 FETCH_SIZE counts half of the bytes of 16-byte-per-lane streaming reads."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def family(name):
    name = re.sub(r"\(.*", "", name)            # drop the argument list
    name = re.sub(r"^void ", "", name)
    return name.strip()


args = sys.argv[1:]
tails = []
while "--tail" in args:    # --tail SUBSTRING N (repeatable): also sum the LAST N dispatches whose kernel name contains SUBSTRING
    i = args.index("--tail")
    tails.append((args[i + 1], int(args[i + 2])))
    del args[i:i + 3]
out = defaultdict(dict)
for d in args:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        rows = list(csv.DictReader(open(path)))
        for row in rows:
            acc[(family(row["Kernel_Name"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
        for tail in tails:
            sel = sorted((r for r in rows if tail[0] in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"]))[-tail[1]:]
            for c in {r["Counter_Name"] for r in sel}:
                v = [float(r["Counter_Value"]) for r in sel if r["Counter_Name"] == c]
                out[f"TAIL last {tail[1]} dispatches of *{tail[0]}*"][c] = {
                    "dispatches": len(v), "mean_KiB": round(sum(v) / len(v), 2), "max_KiB": round(max(v), 2), "sum_KiB": round(sum(v), 1)}
        for (k, c), v in acc.items():
            out[k][c] = {"dispatches": len(v), "mean_KiB": round(sum(v) / len(v), 2), "max_KiB": round(max(v), 2),
                         "sum_KiB": round(sum(v), 1)}
json.dump(dict(sorted(out.items())), sys.stdout, indent=1)
print()
