#!/usr/bin/env python3
"""Concurrency picture of the two-stream step from a rocprofv3 --kernel-trace csv.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --steps 6 --warmup 3 --no-api --no-cpu-baseline
    python tools/kernel_timeline.py gpurun_out/kt/*/*kernel_trace.csv > profiles/<name>.md

Takes the last third of the trace (steady state), and reports: wall time covered by 0 / 1 / 2+ running kernels, per kernel
family the time it ran alone vs beside another kernel, and the gaps between consecutive kernels of the same queue."""
import collections
import csv
import re
import sys


def fam(name):
    n = re.sub(r"\(anonymous namespace\)::|yolo_conv::|void ", "", name)
    m = re.match(r"(\w+)(<[^>]*>)?", n)
    base = m.group(1)
    if base == "conv_igemm_bf16_kernel":
        t = m.group(2).strip("<>").split(",")
        return f"igemm<{t[0].strip()}x{t[1].strip()}>"
    return base[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam(r["Kernel_Name"]), r.get("Queue_Id", "0")) for r in rows]
    ev.sort()
    t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 2 // 3
    ev = [e for e in ev if e[0] >= t_lo]
    pts = []
    for i, (s, e, f, q) in enumerate(ev):
        pts.append((s, 1, i))
        pts.append((e, -1, i))
    pts.sort()
    running = set()
    cover = collections.Counter()
    alone, shared = collections.Counter(), collections.Counter()
    last = pts[0][0]
    for t, d, i in pts:
        dt = t - last
        if dt > 0:
            cover[min(len(running), 2)] += dt
            for j in running:
                (alone if len(running) == 1 else shared)[ev[j][2]] += dt
        last = t
        if d > 0:
            running.add(i)
        else:
            running.discard(i)
    wall = sum(cover.values())
    print(f"window {wall / 1e6:.3f} ms, {len(ev)} kernels\n")
    print("| kernels running | ms | share |\n|---|---|---|")
    for k in (0, 1, 2):
        print(f"| {k if k < 2 else '2+'} | {cover[k] / 1e6:.3f} | {cover[k] / wall:.3f} |")
    print("\n| kernel family | launches | sum of durations ms | alone ms | beside another kernel ms |\n|---|---|---|---|---|")
    cnt = collections.Counter(e[2] for e in ev)
    dur = collections.Counter()
    for s, e, f, q in ev:
        dur[f] += e - s
    for f, d in dur.most_common():
        print(f"| {f} | {cnt[f]} | {d / 1e6:.3f} | {alone[f] / 1e6:.3f} | {shared[f] / 1e6:.3f} |")
    byq = collections.defaultdict(list)
    for s, e, f, q in ev:
        byq[q].append((s, e, f))
    print("\n| queue | kernels | median gap us | mean gap us | gaps > 10 us |\n|---|---|---|---|---|")
    for q, l in byq.items():
        l.sort()
        gaps = [max(0, l[i + 1][0] - l[i][1]) / 1e3 for i in range(len(l) - 1)]
        if not gaps:
            continue
        gs = sorted(gaps)
        print(f"| {q} | {len(l)} | {gs[len(gs) // 2]:.2f} | {sum(gs) / len(gs):.2f} | {sum(g > 10 for g in gs)} |")


if __name__ == "__main__":
    main()
