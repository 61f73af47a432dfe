#!/usr/bin/env python3
"""Per-role timeline of the row-strip inverted-residual kernel (mbstrip_kernel, opt-in form of yolo_mbconv_fwd), diagnostic build.

    python tools/mbstrip_timeline.py --build       # here (CPU): tools/dbg/mbstrip_stamps/libyolo_hip_mbstamps.so =
                                                   # conv_mbconv.hip with -DYOLO_STAMPS + the shipped objects of every other file
    YOLO_HIP_LIB=tools/dbg/mbstrip_stamps/libyolo_hip_mbstamps.so python tools/mbstrip_timeline.py [n,h,w,cin,hidden,cout,stride ...]

The lead wave of every role (expand / depthwise / projection) stamps s_memtime at the top of intervals 4 .. 11 of its band, after the
expand role's stash + fetch, and in front of the interval's barrier.  Printed per block shape, in shader-clock cycles (medians over
workgroups and intervals): the interval length (top to top), each role's busy part (top -> in front of the barrier) and what is
left (its wait at the barrier), with YOLO_MBCONV_DEBUG's role-ablation bits applied as given in the environment."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "dbg", "mbstrip_stamps")
LIB = os.path.join(OUT, "libyolo_hip_mbstamps.so")
sys.path.insert(0, ROOT)


def build():
    from pytorch_yolo_amd import build as B
    B.build()                                            # the shipped objects
    os.makedirs(OUT, exist_ok=True)
    o = os.path.join(OUT, "conv_mbconv_stamps.o")
    subprocess.run([B.HIPCC, *B.COMMON, *B.SOURCES["conv_mbconv.hip"], "-DYOLO_STAMPS", "-c", os.path.join(B.CSRC, "conv_mbconv.hip"), "-o", o], check=True)
    objs = [o if s == "conv_mbconv.hip" else os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES]
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    print(LIB)


def run(spec):
    import numpy as np
    import torch
    from pytorch_yolo_amd import kernels as K
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
    nwg = 4096
    stamps = torch.zeros(nwg, 3, 8, 4, dtype=torch.int64, device=dev)
    os.environ["YOLO_STAMP_PTR"] = hex(stamps.data_ptr())
    for _ in range(3):
        K.mbconv(x, packed, y, **kw)
    torch.cuda.synchronize()
    stamps.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.mbconv(x, packed, y, **kw)
    e1.record()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    live = s[:, 0, 0, 0] > 0
    s = s[live]
    print(f"{spec}: {e0.elapsed_time(e1):.4f} ms, {len(s)} workgroups stamped, YOLO_MBCONV_DEBUG={os.environ.get('YOLO_MBCONV_DEBUG', '')}")
    if not len(s):
        print("  (no stamps: the strip form did not take this block, or the library is not the diagnostic build)")
        return
    top = s[:, :, :, 0].astype(np.float64)
    length = top[:, :, 1:] - top[:, :, :-1]                     # interval k -> k + 1, per role (the roles pass the same barriers)
    print(f"  interval, top to top (cycles)      median {np.median(length[:, 0]):7.0f}   p10 {np.percentile(length[:, 0], 10):7.0f}   p90 {np.percentile(length[:, 0], 90):7.0f}")
    for r, nm in enumerate(["expand role (wave 0)", "depthwise role (wave 6)", "projection role (wave 14)"]):
        busy = (s[:, r, :7, 2] - s[:, r, :7, 0]).astype(np.float64)
        wait = length[:, r] - busy
        line = f"  {nm:26s} busy median {np.median(busy):7.0f}   p90 {np.percentile(busy, 90):7.0f}   barrier wait median {np.median(wait):7.0f}"
        if r == 0:
            sf = (s[:, 0, :7, 1] - s[:, 0, :7, 0]).astype(np.float64)
            line += f"   (stash + fetch {np.median(sf):6.0f})"
        print(line)
    # skew of the tops between roles = how long the barrier release takes to reach each lead wave
    skew = top[:, 1:, :] - top[:, :1, :]
    print(f"  top of interval, depthwise / projection lead wave after the expand lead wave: median {np.median(skew[:, 0]):5.0f} / {np.median(skew[:, 1]):5.0f} cycles")


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    from pytorch_yolo_amd._lib import load
    load().yolo_set_tuning(4, 128 | int(os.environ.get("YOLO_MBCONV_DEBUG", "0")))
    for spec in [a for a in sys.argv[1:] if not a.startswith("--")] or ["64,208,208,32,32,16,1", "64,208,208,16,96,24,2", "64,104,104,24,144,24,1"]:
        run(spec)
