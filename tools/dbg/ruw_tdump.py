"""Debug aid: dump tile 0's intermediate halo (YOLO_RESUNIT_DEBUG=1024) and compare with fp32 torch.  args: n h w c tph"""
import sys
import torch
import torch.nn.functional as F
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import ACT_LEAKY01
DEV = "cuda:0"
n, h, w, c, tph = (int(v) for v in sys.argv[1:6])
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
cmid = c // 2
nimg = cmid // 32
hy_n = 4 * tph + 2
pieces = (hy_n * 22 + 15) // 16
ppw = (pieces + 3) // 4
img_rows = ppw * 4 * 16
raw = y.flatten()[: nimg * img_rows * 32].float().cpu().reshape(nimg, img_rows, 4, 8)      # [image][row][slot][8]
mid = r(F.leaky_relu(F.conv2d(r(x[:1]), r(w1), b1), 0.1))[0]                               # [cmid][h][w]
# tile 0 = image 0, y0 = x0 = 0 (xcd_swizzle(0, grid) = 0)
bad = 0
for hy in range(hy_n):
    for hx in range(22):
        yy, xx = hy - 1, hx - 1
        row = hy * 22 + hx
        for ch8 in range(cmid // 8):
            img, grp = ch8 // 4, ch8 % 4
            slot = grp ^ (2 * (hy & 1))
            got = raw[img, row, slot]
            want = mid[ch8 * 8:ch8 * 8 + 8, yy, xx] if (0 <= yy < h and 0 <= xx < w) else torch.zeros(8)
            if not torch.allclose(got, want, rtol=1e-2, atol=1e-2):
                bad += 1
                if bad <= 12:
                    print("bad hy", hy, "hx", hx, "row", row, "piece", row // 16, "ch8", ch8, "got", got[:4].tolist(), "want", want[:4].tolist())
print("bad groups", bad, "of", hy_n * 22 * (cmid // 8))
