#!/usr/bin/env python3
"""CPU model of where the bf16 path's drift against the fp32 reference comes from (no GPU needed).

The oracle (fp32 restatement of the reference, oracle/models.py) is run three ways (oracle/policy.py) on YOLOv3-SPP with the
bench / golden weights (seed 1234) and image (seed 0):
  fp32            the oracle itself = the reference (bit-equal, tests/test_oracle_golden.py)
  bf16            the HIP fast path's rounding points: BN folded in fp32, conv operands (activations AND folded weights)
                  rounded to bf16, fp32 accumulation, residual sum formed in fp32 and rounded to bf16 once per unit
  bf16_f32stream  the same with the residual stream kept in fp32 (what an fp32 stream buffer would buy)
and the per-layer relative rms error of each policy against fp32 is printed, then the head logit / score / box errors.

    python tests/diag/drift_model.py [--hw 640] [--out profiles/r02_drift_model.md]
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
    t0, res = {}, {}
    io0, p0 = run_policy(om.spp_forward, sd, x, C.SPP_ANCHORS, 80, policy="fp32", taps=t0)
    for pol in ("bf16", "bf16_f32stream"):
        taps = {}
        io, p = run_policy(om.spp_forward, sd, x, C.SPP_ANCHORS, 80, policy=pol, taps=taps)
        res[pol] = (io, p, taps)
    lines = ["| layer | rel rms bf16 | rel rms bf16 + fp32 stream |", "|---|---|---|"]
    for name in t0:
        lines.append(f"| {name} | {rel_rms(res['bf16'][2][name], t0[name]):.5f} | {rel_rms(res['bf16_f32stream'][2][name], t0[name]):.5f} |")
    lines.append("")
    for pol, (io, p, _) in res.items():
        box = (io[..., :4] - io0[..., :4]).abs()
        sc = (io[..., 4:] - io0[..., 4:]).abs()
        lg = [round(float((a - b).abs().max()), 4) for a, b in zip(p, p0)]
        lines.append(f"{pol}: max box err {float(box.max()):.3f} px, max score err {float(sc.max()):.4f}, rms score err "
                     f"{float(sc.double().pow(2).mean().sqrt()):.6f}, max raw-logit err per head {lg}")
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
