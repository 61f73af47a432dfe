"""Time the Darknet stem: fused yolo_stem_fwd vs conv1_nchw + stride-2 conv."""
import sys
import torch
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import ACT_LEAKY01, check, load
import ctypes as C

DEV = "cuda:0"


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    hw = 640
    x = torch.rand(n, 3, hw, hw, device=DEV)
    w1 = torch.randn(32, 3, 3, 3) * 0.27
    w2 = torch.randn(64, 32, 3, 3) * 0.083
    w1p, b1p, kpad1, cpad1 = K.pack_conv_weight(w1, torch.zeros(32), 8)
    w2p, b2p, kpad2, cpad2 = K.pack_conv_weight(w2, torch.zeros(64), 32)
    w1p, b1p, w2p, b2p = (t.to(DEV) for t in (w1p, b1p, w2p, b2p))
    mid = torch.empty(n, hw, hw, 32, dtype=torch.bfloat16, device=DEV)
    y = torch.empty(n, hw // 2, hw // 2, 64, dtype=torch.bfloat16, device=DEV)
    y2 = torch.empty_like(y)
    d1 = K.conv_desc(n=n, h=hw, w=hw, cin=8, in_c_total=8, in_c_offset=0, cout=32, out_c_total=32, out_c_offset=0, ksize=3,
                     stride=1, act=ACT_LEAKY01, kpad=kpad1, cout_pad=cpad1)
    d2 = K.conv_desc(n=n, h=hw, w=hw, cin=32, in_c_total=32, in_c_offset=0, cout=64, out_c_total=64, out_c_offset=0, ksize=3,
                     stride=2, act=ACT_LEAKY01, kpad=kpad2, cout_pad=cpad2)

    def two():
        check(load().yolo_conv1_nchw_f32_fwd(x.data_ptr(), 3, w1p.data_ptr(), b1p.data_ptr(), mid.data_ptr(), C.byref(d1),
                                             K.stream_ptr()), "conv1")
        K.conv2d(mid, w2p, b2p, y2, d2)

    def fused():
        K.stem(x, 3, w1p, b1p, kpad1, w2p, b2p, y, d2)

    tt, tf = bench(two), bench(fused)
    diff = (y.float() - y2.float()).abs()
    print(f"n={n}: fused {tf:.4f} ms   two-kernel {tt:.4f} ms   speedup {tt / tf:.2f}x   max|diff| {float(diff.max()):.4f} "
          f"frac differing {float((diff > 0).float().mean()):.5f}", flush=True)


if __name__ == "__main__":
    main()
