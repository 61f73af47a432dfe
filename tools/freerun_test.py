#!/usr/bin/env python3
"""Experiment: per-step join of the two sub-batch streams vs two free-running pipelines (no join until the end)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from pytorch_yolo_amd import engine, kernels as K
from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
from pytorch_yolo_amd.utils.utils import nms_capacity

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["spp"]
model = wl["cls"](**wl["kw"]).eval()
model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
model = model.to(dev)
x = synth_images(32, 640, 640, 0).to(dev)
S, sub = 2, 16
plans = []
for i in range(S):
    rec = engine.Recorder(sub, 3, 640, 640); model._trace(rec, rec.input)
    plans.append(engine.Plan(rec, dev, 80, 640))
rows, nc = plans[0].rows_total, 80
cap = nms_capacity(rows, nc)
io = torch.empty((32, rows, 85), device=dev)
ps = [plans[0].new_outputs()[1] for _ in range(S)]
dets = torch.empty((32, cap, 7), device=dev); idx = torch.empty((32, cap), dtype=torch.int32, device=dev); cnt = torch.empty((32,), dtype=torch.int32, device=dev)
ws = [torch.empty(K.nms_workspace_bytes(sub, rows, nc), dtype=torch.uint8, device=dev) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]

def pipeline(i):
    lo, hi = i * sub, (i + 1) * sub
    pl = plans[i]
    pl.feed(x[lo:hi]); K.run_ops(pl.op_array, pl.n_ops)
    for hd, p in zip(pl.heads, ps[i]):
        K.decode(hd["sym"].buf.tensor, hd["anchors"], nc, hd["stride"], io[lo:hi], hd["row"], p)
    K.nms_merge(io[lo:hi], 0.1, 0.5, dets[lo:hi], idx[lo:hi], cnt[lo:hi], ws[i])

def step_join():
    cur = torch.cuda.current_stream()
    for i, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s): pipeline(i)
    for s in streams: cur.wait_stream(s)

def step_free():
    for i, s in enumerate(streams):
        with torch.cuda.stream(s): pipeline(i)

for name, fn in (("join per step", step_join), ("free running", step_free), ("join per step", step_join), ("free running", step_free)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per 32 images (full detect pipeline)", flush=True)
