"""Debug aid: where does the fused residual unit differ from fp32 torch?  python tools/dbg/ruw_check.py n h w c"""
import sys
import torch
import torch.nn.functional as F
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import ACT_LEAKY01
DEV = "cuda:0"
n, h, w, c = (int(v) for v in sys.argv[1:5])
g = torch.Generator().manual_seed(c + h)
x = torch.randn(n, c, h, w, generator=g)
w1 = torch.randn(c // 2, c, 1, 1, generator=g) * (2.0 / c) ** 0.5
b1 = torch.randn(c // 2, generator=g) * 0.1
w2 = torch.randn(c, c // 2, 3, 3, generator=g) * (2.0 / (c // 2 * 9)) ** 0.5
b2 = torch.randn(c, generator=g) * 0.1
r = lambda t: t.to(torch.bfloat16).float()
xin = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
y = torch.full((n, h, w, c), -77.0, dtype=torch.bfloat16, device=DEV)
w1p, b1p, kpad1, cpad1 = K.pack_conv_weight(w1, b1, c)
w2p, b2p, kpad2, cpad2 = K.pack_conv_weight(w2, b2, c // 2)
d = K.conv_desc(n=n, h=h, w=w, cin=c // 2, in_c_total=c, in_c_offset=0, cout=c, out_c_total=c, out_c_offset=0, ksize=3, stride=1,
                act=ACT_LEAKY01, kpad=kpad2, cout_pad=cpad2)
K.resunit(xin, w1p.to(DEV), b1p.to(DEV), w2p.to(DEV), b2p.to(DEV), y, d, kpad1, cpad1)
torch.cuda.synchronize()
y2 = torch.full_like(y, -77.0)
K.resunit(xin, w1p.to(DEV), b1p.to(DEV), w2p.to(DEV), b2p.to(DEV), y2, d, kpad1, cpad1)
torch.cuda.synchronize()
print("run-to-run identical:", bool(torch.equal(y, y2)), "differing frac", float((y != y2).float().mean()))
mid = r(F.leaky_relu(F.conv2d(r(x), r(w1), b1), 0.1))
ref = F.leaky_relu(F.conv2d(mid, r(w2), b2, padding=1), 0.1) + r(x)
got = y.float().permute(0, 3, 1, 2).cpu()
err = (got - ref).abs()
bad = err > (2e-2 + 1e-2 * ref.abs())
print("max err", float(err.max()), "bad frac", float(bad.float().mean()))
if bad.any():
    idx = bad.nonzero()
    print("bad per image", torch.bincount(idx[:, 0], minlength=n).tolist())
    print("bad per channel/32", torch.bincount(idx[:, 1] // 32, minlength=c // 32).tolist())
    print("bad per y%8", torch.bincount(idx[:, 2] % 8, minlength=8).tolist(), "y%20", torch.bincount(idx[:, 2] % 20, minlength=20).tolist())
    print("bad per x%20", torch.bincount(idx[:, 3] % 20, minlength=20).tolist())
    print("first", idx[:10].tolist())
