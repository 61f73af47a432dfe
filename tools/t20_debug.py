#!/usr/bin/env python3
"""Where does the 20x20-tile kernel differ from fp32 torch?  Prints the error by pixel patch and by 16-cout fragment."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import ACT_LEAKY01, load

def main():
    n, h, w, cin, cout, knob, use_res = [int(v) for v in (sys.argv[1:] + [1, 20, 20, 32, 128, 16, 0][len(sys.argv) - 1:])]
    lib = load(); dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, h, w, generator=g); wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, bias, cin)
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=cout, out_c_offset=0, ksize=3, stride=1,
                    act=ACT_LEAKY01, kpad=kpad, cout_pad=cout_pad, res=(cout, 0) if use_res else (0, 0))
    res = torch.randn(n, h, w, cout, generator=g).to(torch.bfloat16) if use_res else None
    xin = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    y = torch.zeros(n, h, w, cout, dtype=torch.bfloat16, device=dev)
    old = lib.yolo_set_tuning(2, knob)
    K.conv2d(xin, wp.to(dev), bp.to(dev), y, d, residual=res.to(dev) if use_res else None)
    torch.cuda.synchronize(); lib.yolo_set_tuning(2, old)
    ref = F.leaky_relu(F.conv2d(x.to(torch.bfloat16).float(), wt.to(torch.bfloat16).float(), bias, padding=1), 0.1).permute(0, 2, 3, 1)
    if use_res:
        ref = ref + res.float()
    err = (y.float().cpu() - ref).abs()
    print("max err", float(err.max()))
    e = err[0, :20, :20]
    print("by 4x4 patch (rows = patch row), max over couts:")
    for pr in range(5):
        print("  ", ["%.2f" % float(e[4*pr:4*pr+4, 4*pc:4*pc+4].max()) for pc in range(5)])
    print("by 16-cout fragment:", ["%.2f" % float(e[..., 16*f:16*f+16].max()) for f in range(cout // 16)])
    allbad = (err > 0.1).nonzero()
    print("all bad (n, y%20, x%20, c%64 | y, x, c):", [(int(a), int(b) % 20, int(c) % 20, int(dd) % 64, int(b), int(c), int(dd)) for a, b, c, dd in allbad[:60].tolist()])
    bad = (e > 0.1).nonzero()
    print("first bad (y, x, c):", bad[:12].tolist())
    for yy, xx, cc in bad[:8].tolist():
        print("   got %.4f want %.4f" % (float(y[0, yy, xx, cc]), float(ref[0, yy, xx, cc])))

if __name__ == "__main__":
    main()
