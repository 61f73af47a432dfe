#!/usr/bin/env python3
"""A lone, host-synchronous detect() call: two 16-image halves on two streams (model.detect) against ONE whole-batch list through
the one-call pipeline step (engine.FastStep), same box, interleaved.   python tools/detect_modes.py [workload] [calls]"""
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from pytorch_yolo_amd.utils.synthetic import calibrate_plain_heads, synth_images, synth_state_dict
from pytorch_yolo_amd.utils.utils import nms_capacity


def main():
    wlname = sys.argv[1] if len(sys.argv) > 1 else "spp"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    wl = bench.WORKLOADS[wlname]
    dev = torch.device("cuda", 0)
    model = wl["cls"](**wl["kw"]).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
    model = model.to(dev)
    x = synth_images(wl["bs"], wl["hw"], wl["hw"], 0).to(dev)
    if wlname != "spp":
        calibrate_plain_heads(model, x)
    conf, iou = bench.CONF_THRES, bench.NMS_THRES
    plan = model.plan_for(x)
    io, _ = plan.new_outputs(want_p=False)
    cap = nms_capacity(plan.rows_total, model.n_class)
    out = (torch.empty((wl["bs"], cap, 7), device=dev), torch.empty((wl["bs"], cap), dtype=torch.int32, device=dev),
           torch.empty((wl["bs"],), dtype=torch.int32, device=dev))
    fast = plan.fast_pipeline(0, io, out, conf, iou)

    def whole():
        fast.launch(x)
        return fast.collect()

    with torch.no_grad():
        for _ in range(3):
            a, b = model.detect(x, conf, iou), whole()
        assert all((p is None) == (q is None) for p, q in zip(a, b))
        for rnd in range(3):
            for name, fn in (("detect(): two halves, joined", lambda: model.detect(x, conf, iou)), ("one whole-batch list, FastStep", whole)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                print(f"round {rnd}: {name:36s} {dt * 1e3:.4f} ms per call = {wl['bs'] / dt:.0f} images/s", flush=True)


if __name__ == "__main__":
    main()
