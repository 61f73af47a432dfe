#!/usr/bin/env python3
"""VERDICT r2 item 9, the numerics half, on the CPU: what would Winograd F(2x2, 3x3) on bf16 MFMA operands cost in drift?

The oracle is run under the product's bf16 rounding points (oracle/policy.py, policy 'bf16') and again with every 3x3 / stride-1
convolution replaced by its Winograd form with the roundings an MFMA implementation would have: the input transform B^T d B is
formed in fp32 and ROUNDED TO bf16 (it is the MFMA's B operand), the filter transform G g G^T is formed in fp32 from the folded fp32
weights and rounded to bf16 (the A operand), the 16 element-wise channel contractions accumulate in fp32, the output transform
A^T m A runs in fp32.  Printed: head-logit relative rms, score and box errors against the fp32 oracle for both.

    PYTHONPATH=. python tests/diag/winograd_drift.py [--hw 640]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import blocks as ob                      # noqa: E402
from oracle import models as om                      # noqa: E402
from oracle.policy import bf16r, fold_state_dict     # noqa: E402

BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def winograd_conv3x3(x, w, b):
    """x [n, c, h, w] fp32 (already bf16-rounded activations), w [co, c, 3, 3] fp32 folded weights, pad 1, stride 1."""
    n, c, h, wd = x.shape
    co = w.shape[0]
    ho, wo = h, wd
    th, tw = (ho + 1) // 2, (wo + 1) // 2
    xp = F.pad(x, (1, 1 + 2 * tw - wo, 1, 1 + 2 * th - ho))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                   # [n, c, th, tw, 4, 4]
    v = bf16r(torch.einsum("ij,nctujk,lk->nctuil", BT, d, BT))              # B^T d B, rounded: the MFMA B operand
    u = bf16r(torch.einsum("ij,ocjk,lk->ocil", G, w, G))                    # G g G^T, rounded: the MFMA A operand
    m = torch.einsum("ocil,nctuil->notuil", u, v)                            # fp32 accumulate over channels, per (i, l)
    y = torch.einsum("ij,notujk,lk->notuil", AT, m, AT)                      # [n, co, th, tw, 2, 2]
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(n, co, 2 * th, 2 * tw)[:, :, :ho, :wo]
    return y + b.view(1, -1, 1, 1) if b is not None else y


def run(forward, sd, x, *args, winograd=False, taps=None):
    real = F.conv2d

    def conv(xx, w, b=None, **kw):
        if winograd and w.shape[2] == 3 and kw.get("stride", 1) in (1, (1, 1)) and kw.get("padding", 0) in (1, (1, 1)) and w.shape[1] >= 32:
            return winograd_conv3x3(bf16r(xx), w, b)
        return real(bf16r(xx), bf16r(w), b, **kw)

    def tap(name, t):
        r = bf16r(t) if ".add" in name else None
        if taps is not None:
            taps[name] = t if r is None else r
        return r
    prev = ob.set_tap(tap)
    ob.F.conv2d = conv
    try:
        with torch.no_grad():
            return forward(fold_state_dict(sd), x, *args)
    finally:
        ob.F.conv2d = real
        ob.set_tap(prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=640)
    args = ap.parse_args()
    import _cases as C
    from pytorch_yolo_amd import YOLOv3SPP
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    torch.set_num_threads(os.cpu_count() or 1)
    sd = synth_state_dict(YOLOv3SPP(n_class=80, anchors=C.SPP_ANCHORS).state_dict(), 1234, n_class=80)
    x = synth_images(1, args.hw, args.hw, 0)
    with torch.no_grad():
        io0, p0 = om.spp_forward(sd, x, C.SPP_ANCHORS, 80)
    for name, wg in (("bf16 direct (the shipped rounding points)", False), ("bf16 Winograd F(2x2,3x3) for every 3x3 / s1 layer", True)):
        io, p = run(om.spp_forward, sd, x, C.SPP_ANCHORS, 80, winograd=wg)
        rel = [float((a.double() - b.double()).norm() / b.double().norm()) for a, b in zip(p, p0)]
        sc = (io[..., 4:] - io0[..., 4:]).abs()
        box = (io[..., :4] - io0[..., :4]).abs()
        print(f"{name}: head logits rel rms {[round(r, 5) for r in rel]}, max score err {float(sc.max()):.4f}, rms score err "
              f"{float(sc.double().pow(2).mean().sqrt()):.6f}, max box err {float(box.max()):.2f} px", flush=True)


if __name__ == "__main__":
    main()
