#!/usr/bin/env python3
"""Which rounding points carry the bf16 path's head-logit drift (CPU only, no GPU).

The oracle (fp32 restatement of the reference) is re-run under the product's rounding points (oracle/policy.py: BN folded in fp32, conv
operands bf16, fp32 accumulate, residual stream stored as bf16 once per unit) with the rounding switched on for ONE group of layers at a
time (everything else fp32), and with it switched on everywhere EXCEPT one group - on YOLOv3-SPP 640x640 with the bench / golden weights
(seed 1234) and image (seed 0).  Printed per group: relative rms error of each head's raw logits against fp32, its share of the full
bf16 policy's squared error (independent roundings add in quadrature), and what a mixed mode that keeps that group in fp32 would leave.

    python tests/diag/drift_attribution.py [--hw 640] [--out profiles/r05_drift_attribution.md]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import models as om                     # noqa: E402
from oracle.policy import run_policy                # noqa: E402

GROUPS = [
    ("conv1 + down1 (320x320)", lambda n: n.startswith(("conv1", "down1"))),
    ("down2 (160x160, 2 units)", lambda n: n.startswith("down2")),
    ("down3 (80x80, 8 units)", lambda n: n.startswith("down3")),
    ("down4 (40x40, 8 units)", lambda n: n.startswith("down4")),
    ("down5 (20x20, 4 units)", lambda n: n.startswith("down5")),
    ("neck: sequence_spp + branch1_1", lambda n: n.startswith(("sequence_spp", "branch1_1"))),
    ("branch1_2 (/32 head: 3x3 + head conv)", lambda n: n.startswith("branch1_2")),
    ("branch2_1 + branch2_2 (/16 FPN)", lambda n: n.startswith(("branch2_1", "branch2_2"))),
    ("branch2_3 (/16 head: 3x3 + head conv)", lambda n: n.startswith("branch2_3")),
    ("branch3_1 + branch3_2.conv1-5 (/8 FPN)", lambda n: n.startswith("branch3_1") or (n.startswith("branch3_2") and n[-1] in "12345")),
    ("branch3_2.conv6-7 (/8 head: 3x3 + head conv)", lambda n: n.startswith("branch3_2") and n[-1] in "67"),
    ("all residual-stream stores (.add, 23 units)", lambda n: ".add" in n),
    ("the three head convs only (last 1x1 of each branch)", lambda n: n in ("branch1_2.conv2", "branch2_3.conv7", "branch3_2.conv7")),
]


def rel_rms(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=640)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import _cases as C
    from pytorch_yolo_amd import YOLOv3SPP
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    torch.set_num_threads(8)
    sd = synth_state_dict(YOLOv3SPP(n_class=80, anchors=C.SPP_ANCHORS).state_dict(), 1234, n_class=80)
    x = synth_images(1, args.hw, args.hw, 0)
    io0, p0 = run_policy(om.spp_forward, sd, x, C.SPP_ANCHORS, 80, policy="fp32")

    def drift(select=None):
        io, p = run_policy(om.spp_forward, sd, x, C.SPP_ANCHORS, 80, policy="bf16", select=select)
        return [rel_rms(a, b) for a, b in zip(p, p0)], float((io[..., 4:] - io0[..., 4:]).abs().max())
    full, full_sc = drift()
    lines = [f"YOLOv3-SPP {args.hw}x{args.hw}, weights seed 1234, image seed 0; relative rms error of the raw head logits (heads /32, /16, /8) against the fp32 oracle.",
             f"Full bf16 policy: {full[0]:.5f} / {full[1]:.5f} / {full[2]:.5f}, max score error {full_sc:.4f}.", "",
             "| group of rounding points | ONLY this group in bf16 | share of the full policy's squared error | everything BUT this group in bf16 (mixed mode) | max score error of that mixed mode |",
             "|---|---|---|---|---|"]
    for name, sel in GROUPS:
        # (a stage's selector takes its convs' operand rounding AND its own stream stores: "down3.seq2.1", "down3.add2")
        only, _ = drift(sel)
        rest, rest_sc = drift(lambda n, sel=sel: not sel(n))
        share = [(o / f) ** 2 for o, f in zip(only, full)]
        lines.append(f"| {name} | {only[0]:.5f} / {only[1]:.5f} / {only[2]:.5f} | {share[0]:.2f} / {share[1]:.2f} / {share[2]:.2f} | "
                     f"{rest[0]:.5f} / {rest[1]:.5f} / {rest[2]:.5f} | {rest_sc:.4f} |")
        print(lines[-1], flush=True)
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
