#!/usr/bin/env python3
"""Per-kernel MFMA occupancy from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE --kernel-trace` pass (counter_collection.csv + kernel_trace.csv of the same run).

    python tools/summarize_mfma.py <counter_collection.csv> <kernel_trace.csv>

For every kernel symbol: launches, mean duration, MFMA-pipe busy cycles per launch, mean wave lifetime
(SQ_WAVE_CYCLES counts quad-cycles, MI355X_MICROARCH.md cycle constants), and two clock-independent ratios:
  busy/active = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)  — chip-level MFMA pipe utilisation
  eff. clock  = GRBM_GUI_ACTIVE / 8 / duration (reads high on dispatches < 0.3 ms, DVFS give-back note)
"""
import collections
import csv
import re
import sys


def short(k):
    k = re.sub(r"\(anonymous namespace\)::|yolo_conv::|void ", "", k)
    return k.split("(")[0][:70]


def main():
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))
    nl = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            nl[k] += 1
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(sys.argv[2])):
        dur[short(r["Kernel_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    print("| kernel | launches | us/launch | MFMA busy Mcyc | GUI active/8 kcyc | MFMA busy / (active x 1024 SIMD) | wave life kcyc | eff. clock GHz |")
    print("|---|---|---|---|---|---|---|---|")
    for k, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
        n = nl[k]
        act = c["GRBM_GUI_ACTIVE"] / 8 / n
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n
        life = 4 * c["SQ_WAVE_CYCLES"] / max(1.0, c["SQ_WAVES"])
        d = dur[k] / n
        print(f"| {k} | {n} | {d / 1e3:.1f} | {busy / 1e6:.2f} | {act / 1e3:.1f} | {busy / max(1.0, act * 1024):.3f} | "
              f"{life / 1e3:.1f} | {act / max(1.0, d):.2f} |")


if __name__ == "__main__":
    main()
