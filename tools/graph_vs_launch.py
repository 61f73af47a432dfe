"""Launch list of one sub-batch: direct launches vs HIP-graph replay (one stream)."""
import os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict

def main():
    bs = 16
    wl = bench.WORKLOADS["spp"]
    dev = torch.device("cuda", 0)
    model = wl["cls"](**wl["kw"]).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
    model = model.to(dev); model.n_streams = 1
    x = synth_images(bs, 640, 640, 0).to(dev)
    plan = model.plan_for(x)
    io, ps = plan.new_outputs()
    def t(fn, n=15):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        return statistics.median(ts)
    print("direct launches: %.4f ms" % t(lambda: plan._launch(x, io, ps)))
    print("graph replay   : %.4f ms" % t(lambda: plan.run_graph(x)))
main()
