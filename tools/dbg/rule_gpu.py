"""Per-detection distances of the HIP bf16 path to the reference on the rule cases (tests/_cases.py::RULE_SEEDS)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _cases as C
from helpers import build_rule_case, box_iou
from pytorch_yolo_amd.utils.utils import non_max_suppression
for seed in C.RULE_SEEDS:
    model, sd, x, g = build_rule_case(seed)
    ref = g["nms_dets_0"]
    model = model.to("cuda:0")
    with torch.no_grad():
        io, p = model(x.to("cuda:0"))
        dets, idx = non_max_suppression(io, C.RULE["conf_thres"], C.RULE["nms_thres"], with_indices=True)
    d = dets[0].cpu().numpy()
    print(f"seed {seed}: reference {len(ref)}, bf16 {len(d)}")
    for r in ref:
        mates = [q for q in d if int(q[6]) == int(r[6])]
        best = max(mates, key=lambda q: box_iou(r[:4], q[:4])) if mates else None
        w, h = r[2] - r[0], r[3] - r[1]
        if best is None:
            print(f"   cls {int(r[6]):2d} conf {r[4]:.4f} box {w:5.0f}x{h:5.0f}: NO class mate")
        else:
            print(f"   cls {int(r[6]):2d} conf {r[4]:.4f} (logit {np.log(r[5] / (1 - r[5])):5.2f}) box {w:5.0f}x{h:5.0f}: IoU {box_iou(r[:4], best[:4]):.3f} dconf {best[4] - r[4]:+.4f}")
