"""Time one residual unit: fused yolo_resunit_fwd vs the two-kernel path (1x1 gather + 3x3 halo)."""
import sys
import torch
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import ACT_LEAKY01

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
    for c, hw in ((64, 320), (128, 160), (256, 80)):
        x = torch.randn(n, hw, hw, c, device=DEV).to(torch.bfloat16)
        w1 = torch.randn(c // 2, c, 1, 1) * (2.0 / c) ** 0.5
        w2 = torch.randn(c, c // 2, 3, 3) * (2.0 / (c // 2 * 9)) ** 0.5
        w1p, b1p, kpad1, cpad1 = K.pack_conv_weight(w1, torch.zeros(c // 2), c)
        w2p, b2p, kpad2, cpad2 = K.pack_conv_weight(w2, torch.zeros(c), c // 2)
        w1p, b1p, w2p, b2p = (t.to(DEV) for t in (w1p, b1p, w2p, b2p))
        y = torch.empty_like(x)
        mid = torch.empty(n, hw, hw, c // 2, dtype=torch.bfloat16, device=DEV)
        d = K.conv_desc(n=n, h=hw, w=hw, cin=c // 2, in_c_total=c, in_c_offset=0, cout=c, out_c_total=c, out_c_offset=0,
                        ksize=3, stride=1, act=ACT_LEAKY01, kpad=kpad2, cout_pad=cpad2)
        d1 = K.conv_desc(n=n, h=hw, w=hw, cin=c, in_c_total=c, in_c_offset=0, cout=c // 2, out_c_total=c // 2, out_c_offset=0,
                         ksize=1, stride=1, act=ACT_LEAKY01, kpad=kpad1, cout_pad=cpad1)
        d2 = K.conv_desc(n=n, h=hw, w=hw, cin=c // 2, in_c_total=c // 2, in_c_offset=0, cout=c, out_c_total=c, out_c_offset=0,
                         ksize=3, stride=1, act=ACT_LEAKY01, kpad=kpad2, cout_pad=cpad2, res=(c, 0))

        def fused():
            K.resunit(x, w1p, b1p, w2p, b2p, y, d, kpad1, cpad1)

        def two():
            K.conv2d(x, w1p, b1p, mid, d1)
            K.conv2d(mid, w2p, b2p, x, d2, residual=x)

        flops = 2.0 * n * hw * hw * (c * c // 2 + 9 * c * c // 2)
        tf, tt = bench(fused), bench(two)
        print(f"C={c} {hw}x{hw} n={n}: fused {tf:.4f} ms ({flops / tf / 1e9:.0f} TF/s)   two-kernel {tt:.4f} ms "
              f"({flops / tt / 1e9:.0f} TF/s)   speedup {tt / tf:.2f}x", flush=True)


if __name__ == "__main__":
    main()
