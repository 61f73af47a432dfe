#!/usr/bin/env python3
"""A/B of conv kernels on single layer shapes in ONE process, interleaved rounds (cdna_hip_programming.md 5.4 rule 24).

    python tools/pp_micro.py [--rounds 5] [--reps 20] [--arms 0,4] [n,h,w,cin,cout,k,stride[,res] ...]
Arms are values of the YOLO_CONV_PP knob (yolo_set_tuning(2, v)): 0 = the shipped rules, 4 = ping-pong kernel for every
layer it takes, 12 = also instead of the halo kernel.  Prints median ms / TFLOP/s per arm and the max abs difference of the
outputs against arm 0.
"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import ACT_LEAKY01, load

DEFAULT = ["16,40,40,256,512,3,1,1", "16,20,20,512,1024,3,1,1", "16,80,80,128,256,3,1,1", "16,40,40,512,256,1,1", "16,20,20,1024,512,1,1",
           "16,40,40,768,256,1,1", "16,20,20,2048,512,1,1", "16,20,20,1024,512,3,2", "32,40,40,256,512,3,1,1", "32,20,20,512,1024,3,1,1"]


def main():
    args = sys.argv[1:]
    rounds, reps, arms, knob, half = 5, 20, [0, 4], 2, False
    while args and args[0].startswith("--"):
        if args[0] == "--half":                           # run on a stream that owns half of every XCD's CUs (a pipelined sub-batch stream)
            half = True
            args = args[1:]
            continue
        if args[0] == "--rounds":
            rounds = int(args[1])
        elif args[0] == "--reps":
            reps = int(args[1])
        elif args[0] == "--arms":
            arms = [int(v) for v in args[1].split(",")]
        elif args[0] == "--knob":                         # 2: YOLO_CONV_PP (default), 1: YOLO_CONV_DEBUG, 0: YOLO_CONV_VARIANT
            knob = int(args[1])
        args = args[2:]
    lib = load()
    dev = "cuda:0"
    if half:
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        torch.cuda.synchronize()
        torch.cuda.set_stream(K.cu_masked_stream([b for b in range(n_cu) if (b // 8) < n_cu // 16], torch.device(dev)))
    for spec in args or DEFAULT:
        vals = [int(v) for v in spec.split(",")]
        n, h, w, cin, cout, k, stride = vals[:7]
        use_res = len(vals) > 7 and vals[7]
        pad = (k - 1) // 2
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        g = torch.Generator(device="cpu").manual_seed(1)
        x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).to(dev)
        wt = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, torch.randn(cout, generator=g) * 0.1, cin)
        wp, bp = wp.to(dev), bp.to(dev)
        res = torch.randn(n, ho, wo, cout, generator=g).to(torch.bfloat16).to(dev) if use_res else None
        d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=cout, out_c_offset=0,
                        ksize=k, stride=stride, act=ACT_LEAKY01, kpad=kpad, cout_pad=cout_pad, res=(cout, 0) if use_res else (0, 0))
        ys = {a: torch.zeros(n, ho, wo, cout, dtype=torch.bfloat16, device=dev) for a in arms}
        times = {a: [] for a in arms}
        for a in arms:                                   # warm-up + outputs
            lib.yolo_set_tuning(knob, a)
            for _ in range(3):
                K.conv2d(x, wp, bp, ys[a], d, residual=res)
        torch.cuda.synchronize()
        for _ in range(rounds):
            for a in arms:
                lib.yolo_set_tuning(knob, a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    K.conv2d(x, wp, bp, ys[a], d, residual=res)
                e1.record()
                torch.cuda.synchronize()
                times[a].append(e0.elapsed_time(e1) / reps)
        lib.yolo_set_tuning(knob, 0 if knob else -1)
        fl = 2.0 * n * ho * wo * cout * k * k * cin
        row = f"{spec:28s} M={n * ho * wo:7d} N={cout:5d} K={k * k * cin:5d}"
        for a in arms:
            ms = statistics.median(times[a])
            diff = float((ys[a].float() - ys[arms[0]].float()).abs().max())
            row += f" | pp={a}: {ms:7.4f} ms {fl / ms / 1e9:7.1f} TF (min {min(times[a]):.4f}) maxdiff {diff:.3g}"
        print(row, flush=True)


if __name__ == "__main__":
    main()
