#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference, read-only).  The
reference package cannot be imported as a whole (its __init__ pulls torchvision
and a broken openvino converter), so a bare parent package is
registered and only the hot-path modules are imported:
    pytorch_yolo.models.{yolov3_tiny,yolov3_spp,yolov3,lite_yolo}   pytorch_yolo.utils.utils
``pycocotools`` (used by the eval harness only) is stubbed.

Inputs and weights come from seeded generators (tests/_cases.py,
pytorch_yolo_amd/utils/synthetic.py); the fixtures hold reference OUTPUTS only:

  kat.npz                       SURVEY.md §4 known answers re-captured here
  model_<case>.npz              io, p_k of small models (full tensors) + fused-forward io
  full_<case>.npz               sampled io rows, per-column float64 sums, NMS dets + kept idx
  nms_<case>.npz                dets / kept idx / mutated column 4 per image
  state_keys.json               state_dict key -> shape of the reference models (plain and fused)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")

REF = "/root/reference/pytorch_yolo"


def import_reference():
    pkg = types.ModuleType("pytorch_yolo")
    pkg.__path__ = [REF]
    sys.modules["pytorch_yolo"] = pkg
    for name in ("pycocotools", "pycocotools.cocoeval"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pycocotools.cocoeval"].COCOeval = object
    from pytorch_yolo.models.yolov3_spp import YOLOv3SPP, DownSample
    from pytorch_yolo.models.yolov3_tiny import YOLOv3Tiny
    from pytorch_yolo.models.yolov3 import YOLOv3
    from pytorch_yolo.models.lite_yolo import LiteYOLOv3
    from pytorch_yolo.models.yolo_base import MaxPool
    from pytorch_yolo.utils.utils import non_max_suppression, scale_coords
    return dict(scale_coords=scale_coords, spp=YOLOv3SPP, tiny=YOLOv3Tiny, yolov3=YOLOv3, lite=LiteYOLOv3, MaxPool=MaxPool, DownSample=DownSample,
                nms=non_max_suppression)


def kept_indices(pred_before, dets):
    """Recover, for each output row of the reference NMS, the input row that was
    the pivot of its merge group: conf (col 4) and class are copied unchanged from
    the pivot (utils.py:274), and conf values are unique in the golden inputs."""
    cls = pred_before[:, 5:]
    cpred = np.argmax(cls, 1)
    conf = pred_before[:, 4] * cls[np.arange(len(cls)), cpred]
    out = []
    for row in dets:
        hit = np.nonzero((conf == row[4]) & (cpred == int(row[6])))[0]
        assert hit.size == 1, "golden input has a conf tie — pick another seed"
        out.append(hit[0])
    return np.asarray(out, dtype=np.int64)


def run_ref_nms(ref, pred_np, conf, iou):
    pred = torch.from_numpy(pred_np.copy())
    dets = ref["nms"](pred, conf, iou)
    out = {}
    for b, d in enumerate(dets):
        out[f"col4_{b}"] = pred[b, :, 4].numpy().copy()      # reference mutates it (utils.py:213)
        if d is None:
            out[f"count_{b}"] = np.int64(0)
            continue
        d = d.numpy()
        out[f"count_{b}"] = np.int64(len(d))
        out[f"dets_{b}"] = d
        out[f"kept_{b}"] = kept_indices(pred_np[b], d)
    return out


def main():
    import _cases as C
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict

    torch.set_num_threads(8)
    ref = import_reference()
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None      # regenerate ONE fixture group (separable)

    def save(name, **arrs):
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **arrs)
        print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")

    if only is None:
        _all_but_separable(ref, C, save, synth_images, synth_state_dict)
    _separable_and_nms(ref, C, save, synth_images, synth_state_dict, only)


def _all_but_separable(ref, C, save, synth_images, synth_state_dict):
    # ---- known answers -------------------------------------------------
    mp = ref["MaxPool"](2, 1)(torch.arange(16.).view(1, 1, 4, 4)).numpy()
    pred = torch.from_numpy(C.NMS_KAT_ROWS.copy())[None]
    kat = ref["nms"](pred, **C.NMS_KAT_ARGS)[0].numpy()
    ds = ref["DownSample"](4, 8, repeat=1).eval()
    ds.load_state_dict(synth_state_dict(ds.state_dict(), 5))
    xin = synth_images(1, 16, 16, 6, channels=4)
    with torch.no_grad():
        ds_x, ds_sub = ds(xin)
    sc_in = C.scale_coords_boxes()
    sc = {f"scale_{i}": ref["scale_coords"](s1, torch.from_numpy(sc_in.copy()), s0).numpy()
          for i, (s1, s0) in enumerate(C.SCALE_CASES)}
    save("kat", **sc, maxpool21=mp, nms_kat=kat, nms_kat_col4=pred[0, :, 4].numpy(),
         downsample_x=ds_x.numpy(), downsample_sub=ds_sub.numpy())

    # ---- state_dict key layout (un-fused and fused) the product must reproduce ------------
    import json
    keys = {}
    for fam, kw in (("tiny", dict(kernels_divider=2)), ("spp", dict(kernels_divider=4, anchors=C.SPP_ANCHORS)),
                    ("yolov3", dict(kernels_divider=4, anchors=C.SPP_ANCHORS)), ("lite", dict(kernels_divider=2, anchors=C.SPP_ANCHORS))):
        m = ref[fam](**kw)
        keys[fam] = {k: list(v.shape) for k, v in m.state_dict().items()}
        m.fuse()
        keys[fam + "_fused"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)

    # ---- small models, full tensors ----------------------------------------------------
    for name, (family, kw, bs, h, w, wseed, xseed) in C.MODEL_CASES.items():
        model = ref[family](**kw).eval()
        model.load_state_dict(synth_state_dict(model.state_dict(), wseed, n_class=kw["n_class"]))
        x = synth_images(bs, h, w, xseed)
        with torch.no_grad():
            io, p = model(x)
            model.fuse()
            io_fused, _ = model(x)
        arrs = {"io": io.numpy(), "io_fused": io_fused.numpy()}
        arrs.update({f"p{k}": t.numpy() for k, t in enumerate(p)})
        save("model_" + name, **arrs)

    # ---- BASELINE.json configs at full size: samples + checksums + NMS ------------------
    for name, (family, kw, bs, h, w, wseed, xseed) in C.FULL_CASES.items():
        model = ref[family](**kw).eval()
        model.load_state_dict(synth_state_dict(model.state_dict(), wseed, n_class=kw["n_class"]))
        x = synth_images(bs, h, w, xseed)
        with torch.no_grad():
            io, p = model(x)
        io_np = io.numpy()
        rows = C.sample_rows(io_np.shape[1])
        arrs = {"rows": rows, "io_rows": io_np[:, rows], "io_colsum": io_np.astype(np.float64).sum(1),
                "io_shape": np.asarray(io_np.shape)}
        for k, t in enumerate(p):
            arrs[f"p{k}_sum"] = np.float64(t.double().sum().item())
            arrs[f"p{k}_shape"] = np.asarray(t.shape)
        arrs.update({"nms_" + k: v for k, v in run_ref_nms(ref, io_np, C.NMS_FULL["conf_thres"], C.NMS_FULL["nms_thres"]).items()
                     if not k.startswith("col4")})
        save("full_" + name, **arrs)



def _separable_and_nms(ref, C, save, synth_images, synth_state_dict, only):
    # ---- full-size SPP-640 with well-separated detections (tests/_cases.py, SEPARABLE) -----------
    if only in (None, "separable"):
        sep = C.SEPARABLE
        model = ref["spp"](n_class=80, kernels_divider=1, anchors=C.SPP_ANCHORS).eval()
        sd = synth_state_dict(model.state_dict(), sep["weight_seed"], n_class=80)
        model.load_state_dict(sd)
        x = torch.from_numpy(C.patch_image(sep["image_seed"], sep["n_patches"]))
        with torch.no_grad():
            _, p = model(x)
        wk = [h + ".sequence.batch_norm.weight" for h in C.SEPARABLE_HEADS]
        bk = [h + ".sequence.batch_norm.bias" for h in C.SEPARABLE_HEADS]
        new_w, new_b = C.calibrate_separable_heads([sd[k].numpy() for k in wk], [sd[k].numpy() for k in bk], [t[0].numpy() for t in p],
                                                   80, sep["per_anchor"], sep["span"], sep["gamma_obj"], sep["cls_gain"])
        for k_w, k_b, w_, b_ in zip(wk, bk, new_w, new_b):
            sd[k_w], sd[k_b] = torch.from_numpy(w_), torch.from_numpy(b_)
        model.load_state_dict(sd)
        with torch.no_grad():
            io, p = model(x)
        io_np = io.numpy()
        rows = C.sample_rows(io_np.shape[1])
        arrs = {"rows": rows, "io_rows": io_np[:, rows], "io_colsum": io_np.astype(np.float64).sum(1), "io_shape": np.asarray(io_np.shape)}
        for k, (w_, b_) in enumerate(zip(new_w, new_b)):
            arrs[f"head_bn_weight_{k}"], arrs[f"head_bn_bias_{k}"] = w_, b_       # the calibrated head BN: data derived from reference outputs
        arrs.update({"nms_" + k: v for k, v in run_ref_nms(ref, io_np, sep["conf_thres"], sep["nms_thres"]).items() if not k.startswith("col4")})
        score = io_np[0, :, 4] * io_np[0, :, 5:].max(1)
        arrs["n_candidates"] = np.int64((score > sep["conf_thres"]).sum())
        arrs["n_between_04_06"] = np.int64(((score > 0.4) & (score < 0.6)).sum())
        save("full_spp_640_separable", **arrs)
        print("separable: detections", int(arrs["nms_count_0"]), "candidates", int(arrs["n_candidates"]), "between 0.4 and 0.6:", int(arrs["n_between_04_06"]),
              "min conf", float(arrs["nms_dets_0"][:, 4].min()))
    # ---- full-size SPP-640 cases chosen by a rule on the reference's fp32 outputs alone (tests/_cases.py, RULE / RULE_SEEDS) ----
    if only in (None, "rule"):
        rule = C.RULE
        model = ref["spp"](n_class=80, kernels_divider=1, anchors=C.SPP_ANCHORS).eval()
        sd0 = synth_state_dict(model.state_dict(), rule["weight_seed"], n_class=80)
        for seed in C.RULE_SEEDS:
            model.load_state_dict(sd0)
            x = torch.from_numpy(C.patch_image(seed, rule["n_patches"]))
            with torch.no_grad():
                _, p = model(x)
            sd, gaps = C.rule_state_dict(sd0, [t[0].numpy() for t in p], rule)
            model.load_state_dict(sd)
            with torch.no_grad():
                io, p = model(x)
            io_np = io.numpy()
            ok, why = C.rule_verdict(io_np[0], gaps, rule)
            assert ok, f"rule case seed {seed} does not satisfy the rule on the reference's outputs: {why}"
            rows = C.sample_rows(io_np.shape[1])
            arrs = {"rows": rows, "io_rows": io_np[:, rows], "io_colsum": io_np.astype(np.float64).sum(1), "io_shape": np.asarray(io_np.shape),
                    "cut_gaps": np.asarray([-1.0 if g is None else g for g in gaps])}
            for k, h in enumerate(C.SEPARABLE_HEADS):
                arrs[f"head_bn_weight_{k}"] = sd[h + ".sequence.batch_norm.weight"].numpy()
                arrs[f"head_bn_bias_{k}"] = sd[h + ".sequence.batch_norm.bias"].numpy()
            arrs.update({"nms_" + k: v for k, v in run_ref_nms(ref, io_np, rule["conf_thres"], rule["nms_thres"]).items() if not k.startswith("col4")})
            save(f"full_spp_640_rule_{seed}", **arrs)
            print(f"rule case {seed}: {why}; detections {int(arrs['nms_count_0'])}")
    if only is not None:
        return

    # ---- NMS on synthetic predictions ---------------------------------------------------
    for name in C.NMS_CASES:
        pred_np, conf, iou = C.nms_case_inputs(name)
        save(name, **run_ref_nms(ref, pred_np, conf, iou))


if __name__ == "__main__":
    main()
