#!/usr/bin/env python3
"""Per-kernel sums of the PMC counters of a rocprofv3 run (rocpd sqlite output):  python tools/pmc_summary.py results.db [name filter]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select dispatch_id, kernel_name, counter_name, value, duration, grid_size, workgroup_size from counters_collection").fetchall()
per = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
for did, kn, cn, v, dur, gs, wg in rows:
    if flt and flt not in kn: continue
    per[did][cn] += v; meta[did] = (kn, dur, gs, wg)
by_kernel = collections.defaultdict(list)
for did, c in per.items(): by_kernel[meta[did][0]].append((did, c))
for kn, lst in by_kernel.items():
    lst.sort(); did, c = lst[-1]                      # last dispatch of the kernel (warm)
    print(f"{kn[:90]}  dispatches {len(lst)}  last: duration {meta[did][1]/1000:.1f} us grid {meta[did][2]} wg {meta[did][3]}")
    for k in sorted(c): print(f"    {k:28s} {c[k]:16.0f}")
