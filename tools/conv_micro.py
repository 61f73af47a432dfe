#!/usr/bin/env python3
"""Run single conv layer shapes repeatedly (timing with HIP events; also the target of rocprofv3 --pmc).

    python tools/conv_micro.py [--reps 20] n,h,w,cin,cout,k,stride[,res] ...
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import ACT_LEAKY01


def run(spec, reps):
    vals = [int(v) for v in spec.split(",")]
    n, h, w, cin, cout, k, stride = vals[:7]
    use_res = len(vals) > 7 and vals[7]
    dev = "cuda:0"
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = (torch.randn(n, h, w, cin, device=dev)).to(torch.bfloat16)
    wt = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, torch.zeros(cout), cin)
    wp, bp = wp.to(dev), bp.to(dev)
    y = torch.empty(n, ho, wo, cout, dtype=torch.bfloat16, device=dev)
    res = torch.randn(n, ho, wo, cout, device=dev).to(torch.bfloat16) if use_res else None
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=cout, out_c_offset=0,
                    ksize=k, stride=stride, act=ACT_LEAKY01, kpad=kpad, cout_pad=cout_pad,
                    res=(cout, 0) if use_res else (0, 0))
    for _ in range(3):
        K.conv2d(x, wp, bp, y, d, residual=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        K.conv2d(x, wp, bp, y, d, residual=res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * n * ho * wo * cout * k * k * cin
    print(f"{spec:32s} M={n*ho*wo:8d} N={cout:5d} K={k*k*cin:5d}  {ms:8.4f} ms  {fl/ms/1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    reps = 20
    args = sys.argv[1:]
    if args and args[0] == "--reps":
        reps = int(args[1])
        args = args[2:]
    for s in args or ["32,40,40,256,512,3,1,1", "32,80,80,128,256,3,1,1", "32,80,80,256,128,1,1", "32,20,20,512,1024,3,1,1"]:
        run(s, reps)
