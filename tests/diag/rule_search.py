#!/usr/bin/env python3
"""Seeds for the rule-selected detection-level fixtures (tests/_cases.py::RULE_CASES).

A patch-image seed qualifies by a rule that looks ONLY at the fp32 reference path (here: the fp32 oracle, bit-equal to the reference
on every model golden) - never at a bf16 run, never at the outcome of a comparison:
  R1  per anchor, the objectness cut of calibrate_separable_heads lies in a gap of the normalised conv output >= 4 x 0.007
      (0.007 = the measured bf16 drift of that quantity);
  R2  no row's conf within 0.05 of conf_thres;
  R3  no same-class pair of candidates (conf > conf_thres) with IoU within 0.1 of nms_thres;
  R4  no conf ties among the candidates, and at least 8 of them;
  R5  at every candidate the best class leads the runner-up by >= 2 logits (twice the stated per-logit bound of the bf16 mode).
gamma_obj is moderate (20: the cut does not amplify the drift; 1000 in the round-3 case).  Prints every seed with the rule's verdict;
for the qualifying ones also - for information, NOT for selection - how the bf16-policy re-run of the oracle pairs with fp32.

    PYTHONPATH=. python tests/diag/rule_search.py FIRST LAST [--policy]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _cases as C                                       # noqa: E402
from helpers import strict_share                         # noqa: E402
from oracle import models as om, nms as onms            # noqa: E402
from pytorch_yolo_amd import YOLOv3SPP                  # noqa: E402
from pytorch_yolo_amd.utils.synthetic import synth_state_dict   # noqa: E402


def main():
    torch.set_num_threads(int(os.environ.get("RULE_THREADS", "6")))
    first, last = int(sys.argv[1]), int(sys.argv[2])
    rule = C.RULE
    sd = synth_state_dict(YOLOv3SPP(anchors=C.SPP_ANCHORS).state_dict(), rule["weight_seed"], n_class=80)
    for seed in range(first, last + 1):
        x = torch.from_numpy(C.patch_image(seed, rule["n_patches"]))
        with torch.no_grad():
            _, p = om.spp_forward(sd, x, C.SPP_ANCHORS, 80)
        sd2, gaps = C.rule_state_dict(sd, [t[0].numpy() for t in p], rule)
        with torch.no_grad():
            io, _ = om.spp_forward(sd2, x, C.SPP_ANCHORS, 80)
        ok, why = C.rule_verdict(io.numpy()[0], gaps, rule)
        print(f"seed {seed}: {'QUALIFIES' if ok else 'no'} - {why}", flush=True)
        if ok and "--policy" in sys.argv:
            from oracle.policy import run_policy
            io_b, _ = run_policy(om.spp_forward, sd2, x, C.SPP_ANCHORS, 80, policy="bf16")
            df, _ = onms.non_max_suppression(io.numpy().copy(), rule["conf_thres"], rule["nms_thres"])
            db, _ = onms.non_max_suppression(io_b.numpy().copy(), rule["conf_thres"], rule["nms_thres"])
            print(f"   (information) fp32 {len(df[0])} / bf16-policy {len(db[0])} detections, strict share "
                  f"{strict_share(df[0], db[0]):.3f} / {strict_share(db[0], df[0]):.3f}", flush=True)


if __name__ == "__main__":
    main()
