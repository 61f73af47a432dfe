#!/usr/bin/env python3
"""Experiment: sub-batches small enough that consecutive layers hand activations over through the 256 MB
Infinity Cache.  S streams, each running C chunks back to back (layer list per chunk)."""
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
for S, sub in ((1, 32), (2, 16), (2, 8), (2, 4), (4, 8), (1, 8), (1, 4)):
    plans = []
    for i in range(S):                       # one plan (buffer set) per stream, reused by its chunks
        rec = engine.Recorder(sub, 3, 640, 640); model._trace(rec, rec.input)
        plans.append(engine.Plan(rec, dev, 80, 640))
    streams = [torch.cuda.Stream() for _ in range(S)]
    n_chunks = 32 // sub
    def step():
        cur = torch.cuda.current_stream()
        for s in streams: s.wait_stream(cur)
        for ci in range(n_chunks):
            i = ci % S
            with torch.cuda.stream(streams[i]):
                plans[i].feed(x[ci * sub:(ci + 1) * sub])
                K.run_ops(plans[i].op_array, plans[i].n_ops)
        for s in streams: cur.wait_stream(s)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"streams {S} x sub-batch {sub}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per 32 images", flush=True)
    del plans
    torch.cuda.empty_cache()
