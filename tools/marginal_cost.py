#!/usr/bin/env python3
"""What does a launch family cost INSIDE the pipelined step?  (timing diagnostic; results of the shrunk runs are wrong)

    python tools/marginal_cost.py profiles/r05_layers_spp.txt [--workload spp] > gpurun_out/marginal_cost.txt

The per-layer table (tools/layer_profile.py) times every launch alone on an idle chip; the step runs two lists side by side at the
socket's power cap, where an HBM-bound launch of one pipeline overlaps an MFMA-bound launch of the other.  For each family of launches
this runs `bench.py` with YOLO_SHRINK_OPS=<the family's launch indices> (engine.Plan: those launches process ONE image instead of the
batch - same list, same streams, the launches themselves nearly free) and reports  baseline step - shrunk step  beside the family's
solo sum.  Baselines are taken first, in the middle and last (box drift)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(path):
    out = []
    for ln in open(path):
        m = re.match(r"\s*(\d+)\s+(\w+)\s+(\d+)\s+(\d+)\s+(\d*)\s+(\d)\s+(\d)\s+([\d.]+)\s+([\d.]+)?", ln)
        if m:
            i, kind, M, N, K, k, s, ms = int(m[1]), m[2], int(m[3]), int(m[4]), int(m[5] or 0), int(m[6]), int(m[7]), float(m[8])
            out.append(dict(i=i, kind=kind, M=M, N=N, K=K, k=k, s=s, ms=ms, head="head" in ln))
    return out


def bench(workload, shrink):
    env = dict(os.environ, PYTHONPATH=ROOT)
    if shrink:
        env["YOLO_SHRINK_OPS"] = ",".join(str(i) for i in shrink)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "60", "--warmup", "20", "--no-cpu-baseline",
                        "--no-api", "--no-sustained"], env=env, capture_output=True, text=True, timeout=300)
    j = json.loads(r.stdout.strip().splitlines()[-1])
    return j["ms_per_step"], j["config"].get("mean_detections_per_image")


def main():
    table = rows(sys.argv[1])
    workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "spp"
    mmax = max(r["M"] for r in table if r["kind"] == "conv")
    fam = {}

    def add(name, r):
        fam.setdefault(name, []).append(r)
    for r in table:
        if r["head"]:
            continue
        if "--each" in sys.argv:
            add(f"{r['i']:2d} {r['kind']} M {r['M']} N {r['N']} K {r['K']} k{r['k']} s{r['s']}", r)
        elif r["kind"] not in ("conv", "stem", "resunit"):
            continue
        elif r["kind"] in ("stem", "resunit") or r["M"] >= mmax // 4 and r["i"] < 6:
            add("first stages (stem, fused units, 320->160)", r)
        elif r["s"] == 2:
            add("3x3 / stride 2 (the others)", r)
        else:
            add(f"{r['k']}x{r['k']} on maps of {r['M']} pixels", r)
    base = [bench(workload, None)]
    print(f"baseline {base[0][0]:.4f} ms per step, {base[0][1]} detections per image", flush=True)
    res = []
    names = list(fam)
    for n_, name in enumerate(names):
        if n_ == len(names) // 2:
            base.append(bench(workload, None))
            print(f"baseline {base[-1][0]:.4f} ms per step", flush=True)
        idx = [r["i"] for r in fam[name]]
        ms, dets = bench(workload, idx)
        res.append((name, len(idx), sum(r["ms"] for r in fam[name]), ms, dets))
        print(f"  {name}: {len(idx)} launches shrunk -> {ms:.4f} ms per step ({dets} detections per image)", flush=True)
    base.append(bench(workload, None))
    print(f"baseline {base[-1][0]:.4f} ms per step", flush=True)
    b = sum(x[0] for x in base) / len(base)
    print(f"\nbaseline mean {b:.4f} ms ({', '.join(f'{x[0]:.4f}' for x in base)})\n")
    print("| family | launches | solo sum (ms) | step with the family shrunk to one image (ms) | marginal cost in the step (ms) | marginal / solo |")
    print("|---|---|---|---|---|---|")
    tot_solo = tot_marg = 0.0
    for name, n, solo, ms, dets in res:
        print(f"| {name} | {n} | {solo:.3f} | {ms:.4f} | {b - ms:.3f} | {(b - ms) / solo:.2f} |")
        tot_solo += solo
        tot_marg += b - ms
    print(f"| all of the above | | {tot_solo:.3f} | | {tot_marg:.3f} | {tot_marg / tot_solo:.2f} |")


if __name__ == "__main__":
    main()
