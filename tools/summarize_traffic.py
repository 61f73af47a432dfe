#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) per kernel family.

    python tools/summarize_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <steps> > profiles/<name>.md

gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half the bytes of wide coalesced reads
-> doubled here; WRITE_SIZE is exact for 16-byte-per-lane stores.  Counter unit: KiB.
"""
import collections
import csv
import re
import sys


def load(path, counter):
    per, n = collections.OrderedDict(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"yolo_conv::", "", k)
        k = ("conv kernels (conv_igemm_bf16 incl. head+decode / conv3x3_t20 / conv1x1_stream / conv3x3_halo / resunit / stem)"
             if re.search(r"conv_igemm|conv3x3_halo|conv3x3_t20|conv3x3s2_t20|conv1x1_stream|conv1_nchw|resunit\w*_kernel|stem2?_kernel", k) else k.split("(")[0][:60])
        per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
        n[k] += 1
    return per, n


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    steps = float(sys.argv[3])   # 32-image steps profiled = stem_kernel launches / 2 (two sub-batch streams)
    print("| kernel | launches/step | FETCH_SIZE raw MB/step | read MB/step (x2, gfx950) | WRITE_SIZE MB/step | HBM MB/step |")
    print("|---|---|---|---|---|---|")
    for k in fetch:
        f, w = fetch[k] / 1024 / steps, write.get(k, 0.0) / 1024 / steps
        print(f"| {k} | {nf[k] / steps:.0f} | {f:.1f} | {2 * f:.1f} | {w:.1f} | {2 * f + w:.1f} |")


if __name__ == "__main__":
    main()
