"""Standalone time of each stage of detect() for one sub-batch on one stream (HIP events, median)."""
import os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
from pytorch_yolo_amd.utils.utils import nms_capacity, nms_launch

def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    wl = bench.WORKLOADS["spp"]
    dev = torch.device("cuda", 0)
    model = wl["cls"](**wl["kw"]).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
    model = model.to(dev); model.n_streams = 1
    x = synth_images(bs, 640, 640, 0).to(dev)
    plan = model.plan_for(x)
    io, ps = plan.new_outputs()
    cap = nms_capacity(plan.rows_total, 80)
    out = (torch.empty((bs, cap, 7), device=dev), torch.empty((bs, cap), dtype=torch.int32, device=dev), torch.empty((bs,), dtype=torch.int32, device=dev))
    stages = {
        "layer list": lambda: (plan.feed(x), plan._bind_outputs(io, ps), K.run_ops(plan.op_array, plan.n_ops)),
        "decode (unfused heads)": lambda: plan._decode_unfused(io, ps),
        "nms": lambda: nms_launch(io, 0.1, 0.5, out, slot=0),
    }
    for _ in range(3):
        for f in stages.values(): f()
    torch.cuda.synchronize()
    for name, f in stages.items():
        ts = []
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"bs={bs} {name:18s} {statistics.median(ts):.4f} ms", flush=True)

main()
