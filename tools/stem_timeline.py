#!/usr/bin/env python3
"""Step timeline of the role-split stem kernel (stem2_kernel), diagnostic build with stamps.

    python tools/block_timeline.py --build          # here: tools/_stamps/libyolo_hip_stamps.so (-DYOLO_STAMPS)
    YOLO_HIP_LIB=tools/_stamps/libyolo_hip_stamps.so python tools/stem_timeline.py [n]      # on the GPU box

Wave 0 (producer) and wave 4 (consumer) of every workgroup stamp s_memrealtime (100 MHz) in steps 8..23.  Producer: 0 step start,
2 halo committed to LDS + next loads issued, 1 conv1 done, 3 LDS drained (the barrier follows), 4..8 conv1 blocks.  Consumer: 0 step start, 1 taps + epilogue
issued, 3 LDS drained.  Printed: medians of every phase and of the barrier waits."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_yolo_amd import kernels as K          # noqa: E402
from pytorch_yolo_amd._lib import ACT_LEAKY01      # noqa: E402

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hw = 640
x = torch.rand(n, 3, hw, hw, device=DEV)
w1p, b1p, kpad1, _ = K.pack_conv_weight(torch.randn(32, 3, 3, 3) * 0.27, torch.zeros(32), 8)
w2p, b2p, kpad2, cpad2 = K.pack_conv_weight(torch.randn(64, 32, 3, 3) * 0.083, torch.zeros(64), 32)
w1p, b1p, w2p, b2p = (t.to(DEV) for t in (w1p, b1p, w2p, b2p))
y = torch.empty(n, hw // 2, hw // 2, 64, dtype=torch.bfloat16, device=DEV)
d = K.conv_desc(n=n, h=hw, w=hw, cin=32, in_c_total=32, in_c_offset=0, cout=64, out_c_total=64, out_c_offset=0, ksize=3, stride=2,
                act=ACT_LEAKY01, kpad=kpad2, cout_pad=cpad2)
stamps = torch.zeros(256, 2, 16, 12, dtype=torch.int64, device=DEV)
os.environ["YOLO_STAMP_PTR"] = hex(stamps.data_ptr())
for _ in range(5):
    K.stem(x, 3, w1p, b1p, kpad1, w2p, b2p, y, d)
torch.cuda.synchronize()
stamps.zero_()
torch.cuda.synchronize()
K.stem(x, 3, w1p, b1p, kpad1, w2p, b2p, y, d)
torch.cuda.synchronize()
raw = stamps.cpu().numpy()
s = raw.astype(np.float64) / 100.0          # us
okr = raw[:, 0, 0, 0] > 0
clk = (raw[okr, 0, 15, 10] - raw[okr, 0, 0, 10]) / ((raw[okr, 0, 15, 0] - raw[okr, 0, 0, 0]) / 100.0)      # s_memtime ticks per us
print(f"shader clock over steps 8..23: median {np.median(clk):.0f} MHz (s_memtime ticks per s_memrealtime microsecond)")
ok = s[:, 0, 0, 0] > 0
s = s[ok]
print(f"{len(s)} workgroups stamped, {n} images")
p, c = s[:, 0], s[:, 1]                                       # [wg, step, point]


def med(a):
    return f"{np.median(a):6.3f} us (p10 {np.percentile(a, 10):6.3f}, p90 {np.percentile(a, 90):6.3f})"


steps = slice(1, 15)
print("step length (producer start to next start)", med(p[:, 2:16, 0] - p[:, 1:15, 0]))
print("producer: commit + fetch   ", med((p[:, :, 2] - p[:, :, 0])[:, steps]))
print("producer: conv1            ", med((p[:, :, 1] - p[:, :, 2])[:, steps]))
print("producer: LDS drain        ", med((p[:, :, 3] - p[:, :, 1])[:, steps]))
print("producer: barrier wait     ", med(p[:, 2:16, 0] - p[:, 1:15, 3]))
print("consumer: taps + epilogue  ", med((c[:, :, 1] - c[:, :, 0])[:, steps]))
print("consumer: LDS drain        ", med((c[:, :, 3] - c[:, :, 1])[:, steps]))
print("consumer: barrier wait     ", med(c[:, 2:16, 0] - c[:, 1:15, 3]))
for k in range(4, 9):
    print(f"producer: conv1 block {k - 4} start, since step start", med((p[:, :, k] - p[:, :, 0])[:, steps]))
for k in range(4, 8):
    print(f"consumer: tap {3 * (k - 4)} start / taps end, since step start", med((c[:, :, k] - c[:, :, 0])[:, steps]))
