#!/usr/bin/env python
"""Shorten the kernel names of a rocprofv3 *_kernel_stats.csv so the summary is readable and
small enough to commit under profiles/.  Usage: trim_stats.py in.csv out.csv [max_rows]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^()]{0,60}>)?)", name)
    s = m.group(1) if m else name
    return s[:100]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:1 + limit]:
            r[0] = short(r[0])
            w.writerow(r)


if __name__ == "__main__":
    main()
