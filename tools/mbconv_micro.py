#!/usr/bin/env python3
"""Time yolo_mbconv_fwd on single block shapes (HIP events).

    python tools/mbconv_micro.py n,h,w,cin,hidden,cout,stride ...
YOLO_MBCONV_DEBUG bits 2 / 4 / 8 / 16 drop the expand / depthwise / projection stage / the x loads (timing only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_yolo_amd import kernels as K


def run(spec, reps=20):
    n, h, w, cin, hidden, cout, stride = [int(v) for v in spec.split(",")]
    dev = "cuda:0"
    has_exp, has_res = hidden != cin, stride == 1 and cin == cout
    x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
    we = torch.randn(hidden, cin, 1, 1) * (2.0 / cin) ** 0.5 if has_exp else None
    be = torch.randn(hidden) * 0.5 if has_exp else None
    packed = tuple(None if t is None else t.to(dev) for t in K.pack_mbconv(
        we, be, torch.randn(hidden, 1, 3, 3) * 0.4, torch.randn(hidden) * 0.5, torch.randn(cout, hidden, 1, 1) * hidden ** -0.5, torch.randn(cout) * 0.1))
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.empty(n, ho, wo, cout, dtype=torch.bfloat16, device=dev)
    kw = dict(n=n, h=h, w=w, cin=cin, hidden=hidden, cout=cout, in_view=(cin, 0), out_view=(cout, 0), stride=stride, has_res=has_res)
    for _ in range(3):
        K.mbconv(x, packed, y, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        K.mbconv(x, packed, y, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    by = x.numel() * 2 + y.numel() * 2
    print(f"{spec:28s} {ms:8.4f} ms  {by / ms / 1e6:7.0f} GB/s (x + y)", flush=True)


if __name__ == "__main__":
    for s in sys.argv[1:] or ["64,208,208,32,32,16,1", "64,208,208,16,96,24,2", "64,104,104,24,144,24,1", "64,104,104,24,144,32,2",
                              "64,52,52,32,192,32,1", "64,52,52,32,192,64,2"]:
        run(s)
