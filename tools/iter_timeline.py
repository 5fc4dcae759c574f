#!/usr/bin/env python
"""One iteration of tools/decoder_bench.py as a timeline: kernel, duration, idle gap before it -- from a rocprofv3
--kernel-trace CSV.  usage: iter_timeline.py kernel_trace.csv [anchor kernel substring = project_fwd_kernel]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "project_fwd_kernel"
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
idx = [i for i, k in enumerate(ks) if anchor in k[2]]
a, b = idx[-3], idx[-2]  # a full iteration in the middle of the timed loop
it = ks[a:b]
busy = sum(e - s for s, e, _ in it)
span = ks[b][0] - ks[a][0]
print(f"iteration: {span / 1e6:.3f} ms wall, {busy / 1e6:.3f} ms in {len(it)} kernels, {(span - busy) / 1e6:.3f} ms idle")
prev = it[0][0]
for s, e, n in it:
    print(f"{(s - it[0][0]) / 1e3:9.1f} us  +{(s - prev) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {n[:90]}")
    prev = max(prev, e)
