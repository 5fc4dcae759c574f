#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes: per kernel (short name) mean counter value per dispatch.

With a third argument, also writes the per-launch HBM traffic of every kernel as JSON (bench.py reads it for
`roofline.traffic`): fetch = 2 * FETCH_SIZE KiB (gfx950: FETCH_SIZE tallies 128-B requests at 64 B,
MI355X_MICROARCH.md "HBM"), write = WRITE_SIZE KiB (checked here against a known byte count: the rows kernel
stores exactly rows * D * 4 bytes and WRITE_SIZE reports that number)."""
import json
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^()]{0,40}>)?)", name)
    return (m.group(1) if m else name)[:60]


def main():
    root = sys.argv[1]
    want = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"raster_|rs_|project|tile_|scan_")
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(dict)
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if not want.search(k):
                continue
            per_dispatch[(k, row["Dispatch_Id"])][row["Counter_Name"]] = \
                per_dispatch[(k, row["Dispatch_Id"])].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        for (k, _), ctrs in per_dispatch.items():
            for c, v in ctrs.items():
                acc[k][c].append(v)
    if len(sys.argv) > 3:
        traffic = {}
        for k in acc:
            if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
                f = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"]) * 1024.0 * 2.0
                w = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"]) * 1024.0
                traffic[k] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w}
                for key, ctr in (("mfma_insts", "SQ_INSTS_MFMA"), ("mfma_busy_cycles", "SQ_VALU_MFMA_BUSY_CYCLES"),
                                 ("busy_cu_cycles", "SQ_BUSY_CU_CYCLES"), ("valu_insts", "SQ_INSTS_VALU"),
                                 ("lds_bank_conflict", "SQ_LDS_BANK_CONFLICT"), ("wave_cycles", "SQ_WAVE_CYCLES"),
                                 ("wait_inst_any", "SQ_WAIT_INST_ANY")):
                    if ctr in acc[k]:
                        traffic[k][key] = sum(acc[k][ctr]) / len(acc[k][ctr])
        json.dump(traffic, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            vals = acc[k][c]
            print(f"    {c:34s} mean/dispatch {sum(vals) / len(vals):18.1f}   (n={len(vals)})")


if __name__ == "__main__":
    main()
