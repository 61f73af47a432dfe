#!/usr/bin/env python3
"""Does splitting the batch over S concurrent HIP streams hide the tile-quantisation tail?  (experiment)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from pytorch_yolo_amd import engine, kernels as K
from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["spp"]
model = wl["cls"](**wl["kw"]).eval()
model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
model = model.to(dev)
x = synth_images(32, 640, 640, 0).to(dev)
for S in (1, 2, 4):
    sub = 32 // S
    plans, xs = [], []
    for i in range(S):
        rec = engine.Recorder(sub, 3, 640, 640); model._trace(rec, rec.input)
        plans.append(engine.Plan(rec, dev, 80, 640)); xs.append(x[i * sub:(i + 1) * sub].contiguous())
    streams = [torch.cuda.Stream() for _ in range(S)]
    def step():
        cur = torch.cuda.current_stream()
        for s in streams: s.wait_stream(cur)
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                K.pack_input(xs[i], plans[i].input_buffer)
                K.run_ops(plans[i].op_array, plans[i].n_ops)
        for s in streams: cur.wait_stream(s)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"streams {S}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per 32 images (pack + conv layers)", flush=True)
