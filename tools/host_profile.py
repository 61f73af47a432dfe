#!/usr/bin/env python3
"""Where the host time of the API calls goes: cProfile over model.detect() / model.detect_stream().

    python tools/host_profile.py spp detect 20
    python tools/host_profile.py tiny stream 300
"""
import cProfile
import importlib.util
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from pytorch_yolo_amd.utils.synthetic import calibrate_plain_heads, synth_images, synth_state_dict


def main():
    wlname, mode, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    wl = bench.WORKLOADS[wlname]
    dev = torch.device("cuda", 0)
    model = wl["cls"](**wl["kw"]).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
    model = model.to(dev)
    x = synth_images(wl["bs"], wl["hw"], wl["hw"], 0).to(dev)
    if wlname != "spp":
        calibrate_plain_heads(model, x)
    conf, iou = bench.CONF_THRES, bench.NMS_THRES
    with torch.no_grad():
        if mode == "detect":
            run = lambda k: [model.detect(x, conf, iou) for _ in range(k)]
        else:
            run = lambda k: [r for r in model.detect_stream((x for _ in range(k)), conf, iou)]
        run(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{wlname} {mode}: {n} batches of {wl['bs']} in {dt * 1e3:.2f} ms = {dt / n * 1e3:.4f} ms per batch = {wl['bs'] * n / dt:.0f} images/s")
        pr = cProfile.Profile()
        pr.enable()
        run(n)
        torch.cuda.synchronize()
        pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
