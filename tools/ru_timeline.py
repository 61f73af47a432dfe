#!/usr/bin/env python3
"""Phase timeline of the fused residual-unit kernel (resunit_t20w_kernel), diagnostic build with stamps.

    python tools/block_timeline.py --build          # here: tools/_stamps/libyolo_hip_stamps.so (-DYOLO_STAMPS)
    YOLO_HIP_LIB=tools/_stamps/libyolo_hip_stamps.so python tools/ru_timeline.py C HW [n]      # on the GPU box

Wave 0 of every workgroup stamps s_memrealtime (100 MHz) at: 0 start, 1 first half of x landed, 2 x in registers, 3 t written
(barrier passed), 4 nine taps done, 5 stores issued and drained.  Printed: the median length of every phase, how the phases of
the two workgroups of a CU overlap, and the launch's span."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_yolo_amd import kernels as K          # noqa: E402
from pytorch_yolo_amd._lib import ACT_LEAKY01      # noqa: E402

DEV = "cuda:0"
c, hw = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
x = torch.randn(n, hw, hw, c, device=DEV).to(torch.bfloat16)
w1 = torch.randn(c // 2, c, 1, 1) * (2.0 / c) ** 0.5
w2 = torch.randn(c, c // 2, 3, 3) * (2.0 / (c // 2 * 9)) ** 0.5
w1p, b1p, kpad1, cpad1 = K.pack_conv_weight(w1, torch.zeros(c // 2), c)
w2p, b2p, kpad2, cpad2 = K.pack_conv_weight(w2, torch.zeros(c), c // 2)
w1p, b1p, w2p, b2p = (t.to(DEV) for t in (w1p, b1p, w2p, b2p))
y = torch.empty_like(x)
d = K.conv_desc(n=n, h=hw, w=hw, cin=c // 2, in_c_total=c, in_c_offset=0, cout=c, out_c_total=c, out_c_offset=0, ksize=3, stride=1,
                act=ACT_LEAKY01, kpad=kpad2, cout_pad=cpad2)
stamps = torch.zeros(1 << 15, 8, dtype=torch.int64, device=DEV)
os.environ["YOLO_STAMP_PTR"] = hex(stamps.data_ptr())
for _ in range(5):
    K.resunit(x, w1p, b1p, w2p, b2p, y, d, kpad1, cpad1)
torch.cuda.synchronize()
stamps.zero_()
torch.cuda.synchronize()
K.resunit(x, w1p, b1p, w2p, b2p, y, d, kpad1, cpad1)
torch.cuda.synchronize()
s = stamps.cpu().numpy()
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
if c == 64:      # resunit64_t20_kernel: 0 start, 1 halo landed, 2 intermediate written + barrier, 3 nine taps done, 4 epilogue done
    us = (s[:, :5] - s[:, :1]) / 100.0
    names = ["halo DMA (issue + wait)", "1x1 + intermediate + barrier", "nine taps", "epilogue"]
    print(f"C=64 {hw}x{hw} n={n}: {len(s)} workgroups")
    for k, nm in enumerate(names):
        dur = us[:, k + 1] - us[:, k]
        print(f"  {nm:30s} median {np.median(dur):6.2f} us   p10 {np.percentile(dur, 10):6.2f}   p90 {np.percentile(dur, 90):6.2f}")
    print(f"  whole workgroup                median {np.median(us[:, 4]):6.2f} us")
    sys.exit(0)
us = (s[:, :6] - t0) / 100.0
names = ["first x chunk", "x chunks x W1", "t write", "nine taps", "epilogue"]
print(f"C={c} {hw}x{hw} n={n}: {len(s)} workgroups, span {us[:, 5].max():.1f} us")
for k, nm in enumerate(names):
    dur = us[:, k + 1] - us[:, k]
    print(f"  {nm:18s} median {np.median(dur):6.2f} us   p10 {np.percentile(dur, 10):6.2f}   p90 {np.percentile(dur, 90):6.2f}")
print(f"  whole workgroup    median {np.median(us[:, 5] - us[:, 0]):6.2f} us")
cu = s[:, 6]
key = ((cu >> 32) & 0xf) * 1000 + ((cu >> 13) & 0x7) * 100 + ((cu >> 8) & 0xf)        # XCC, SE, CU
print("  distinct CUs", len(np.unique(key)), " starts in the first 2 us:", int((us[:, 0] < 2.0).sum()))
# overlap: for every workgroup, which phase was the OTHER resident workgroup of its CU in when this one began its nine taps
order = np.argsort(us[:, 0])
hist = {}
for k_ in np.unique(key):
    idx = np.where(key == k_)[0]
    for i in idx:
        tb = us[i, 3]
        for j in idx:
            if j != i and us[j, 0] <= tb < us[j, 5]:
                ph = int(np.searchsorted(us[j, 1:6], tb, side="right"))
                hist[names[min(ph, 4)]] = hist.get(names[min(ph, 4)], 0) + 1
print("  phase of the co-resident workgroup when a workgroup starts its nine taps:", hist)
first = us[order[:8], 0]
print("  first starts (us):", np.round(first, 2).tolist(), " start of workgroups 256..263 (grid order):", np.round(us[256:264, 0], 2).tolist() if len(us) > 264 else "")
