#!/usr/bin/env python3
"""Counter traffic per kernel AND grid (= per layer shape) from the two rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE).

    python tools/traffic_by_kernel.py <fetch counter_collection.csv> <write counter_collection.csv> <passes> [min MB/step]

`passes` = whole-batch lists the run executed (tools/summarize_traffic.py's `steps`).  Same corrections as summarize_traffic.py:
counter unit KiB, FETCH_SIZE doubled on gfx950 for wide coalesced reads.  Output: markdown, largest first.
"""
import collections
import csv
import re
import sys


def load(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(anonymous namespace\)::|yolo_conv::", "", r["Kernel_Name"])
        k = re.sub(r"^void ", "", k).split("(")[0]
        key = (k[:70], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        tot[key] += float(r["Counter_Value"])
        n[key] += 1
    return tot, n


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    passes = float(sys.argv[3])
    floor_mb = float(sys.argv[4]) if len(sys.argv) > 4 else 50.0
    rows = []
    for k in fetch:
        rd, wr = 2 * fetch[k] / 1024 / passes, write.get(k, 0.0) / 1024 / passes
        rows.append((rd + wr, k, nf[k] / passes, rd, wr))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print("| kernel | workgroups | launches / list | read MB / list | written MB / list | MB / list | MB / launch |")
    print("|---|---|---|---|---|---|---|")
    for t, (k, g), n, rd, wr in rows:
        if t >= floor_mb:
            print(f"| `{k}` | {g} | {n:.0f} | {rd:.1f} | {wr:.1f} | {t:.1f} | {t / n:.1f} |")
    print(f"\nall kernels: {total:.1f} MB per list")


if __name__ == "__main__":
    main()
