#!/usr/bin/env python3
"""How much of every `nms_merge` launch runs beside a layer kernel of a launch list?

Reads a `rocprofv3 --kernel-trace` CSV of `bench.py` (tools/r5_profiles.sh, the `kstats` passes).  The NMS of a batch runs on a
side stream while the other pipeline's list (and the next batch's list) runs on the pipeline streams; a step is shorter by the
NMS's duration only if the NMS would otherwise sit on the critical path.  Per `nms_merge*` dispatch: the part of [start, end]
covered by the union of all OTHER kernels' intervals.

  python tools/nms_overlap.py gpurun_out/r5p_kstats_spp/runc/*_kernel_trace.csv
"""
import csv
import sys


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]))
    nms = [(s, e) for n, s, e, _ in rows if "nms_merge_kernel" in n]
    other = sorted((s, e) for n, s, e, _ in rows if "nms_" not in n and "pack_detections" not in n)
    # union of the other kernels' intervals
    merged = []
    for s, e in other:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    import bisect
    starts = [m[0] for m in merged]
    tot = cov = 0
    fully = 0
    for s, e in nms:
        c = 0
        i = max(0, bisect.bisect_right(starts, s) - 1)
        while i < len(merged) and merged[i][0] < e:
            c += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
            i += 1
        tot += e - s
        cov += c
        fully += c >= 0.999 * (e - s)
    n = len(nms)
    print(f"{path}")
    print(f"  nms_merge dispatches {n}, average {tot / n / 1e3:.1f} us; covered by a layer kernel of some list: {cov / tot:.3f} of their time; "
          f"{fully} of {n} dispatches covered from start to end")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
