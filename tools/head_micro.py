"""Time the fused head conv + decode on the three SPP-640 head shapes."""
import sys, torch
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import ACT_LEAKY01, DT_F32
DEV = "cuda:0"
def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for hw, cin in ((80, 256), (40, 512), (20, 1024)):
        nc, na = 80, 3
        x = torch.randn(n, hw, hw, cin, device=DEV).to(torch.bfloat16)
        wt = torch.randn(255, cin, 1, 1) * (1.0 / cin) ** 0.5
        wp, bp, kpad, cpad = K.pack_conv_weight(wt, torch.zeros(255), cin)
        wp, bp = wp.to(DEV), bp.to(DEV)
        d = K.conv_desc(n=n, h=hw, w=hw, cin=cin, in_c_total=cin, in_c_offset=0, cout=255, out_c_total=256, out_c_offset=0,
                        ksize=1, stride=1, act=ACT_LEAKY01, kpad=kpad, cout_pad=cpad, out_dtype=DT_F32)
        io = torch.empty(n, na * hw * hw, 85, device=DEV)
        p = torch.empty(n, na, hw, hw, 85, device=DEV)
        f = lambda: K.head_decode(x, wp, bp, d, [(10., 13.), (33., 23.), (59., 119.)], nc, 640 / hw, io, 0, p)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print(f"head {hw}x{hw} cin {cin} n={n}: {ms:.4f} ms  ({(io.numel() + p.numel()) * 4 / ms / 1e6:.0f} GB/s written)", flush=True)
main()
