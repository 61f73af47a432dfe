#!/usr/bin/env python3
"""Socket power and shader clock while ONE conv layer runs back to back, on the whole chip or on half of every XCD's CUs.

    python tools/power_probe.py [--half] [--seconds 6] n,h,w,cin,cout,k,stride[,res]

Launches the layer in a loop for the given time and samples `rocm-smi --showpower --showclocks` from a helper process
twice a second; prints the layer's average time and the samples.  (Evidence for DESIGN.md 3.2a: the MFMA-dense layers run
against the socket power limit, not against an issue or memory limit.)"""
import os
import re
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import ACT_LEAKY01


def main():
    args = sys.argv[1:]
    half, seconds = False, 6.0
    while args and args[0].startswith("--"):
        if args[0] == "--half":
            half = True
            args = args[1:]
        elif args[0] == "--seconds":
            seconds = float(args[1])
            args = args[2:]
    vals = [int(v) for v in args[0].split(",")]
    n, h, w, cin, cout, k, stride = vals[:7]
    use_res = len(vals) > 7 and vals[7]
    dev = torch.device("cuda", 0)
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
    y = torch.zeros(n, ho, wo, cout, dtype=torch.bfloat16, device=dev)
    if half:
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        torch.cuda.synchronize()
        torch.cuda.set_stream(K.cu_masked_stream([b for b in range(n_cu) if (b // 8) < n_cu // 16], dev))
    for _ in range(20):
        K.conv2d(x, wp, bp, y, d, residual=res)
    torch.cuda.synchronize()
    samples = []
    t0 = time.perf_counter()
    iters = 0
    next_sample = t0 + 1.0
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            K.conv2d(x, wp, bp, y, d, residual=res)
        iters += 200
        if time.perf_counter() >= next_sample:             # the queue is a few ms deep: the card is busy while rocm-smi runs
            out = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
            pw = re.search(r"Power \(W\): ([\d.]+)", out)
            sc = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            samples.append((float(pw.group(1)) if pw else -1.0, int(sc.group(1)) if sc else -1))
            next_sample = time.perf_counter() + 0.5
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fl = 2.0 * n * ho * wo * cout * k * k * cin
    print(f"{args[0]} {'half' if half else 'whole'} chip: {dt / iters * 1e3:.4f} ms/launch incl. sampling pauses, {fl * iters / dt / 1e12:.0f} TFLOP/s; "
          f"(W, sclk MHz) samples: {samples}")


if __name__ == "__main__":
    main()
