"""CPU: the oracle (oracle/) against the golden vectors captured from the reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import _cases as C
from helpers import build_case, build_rule_case, build_separable_case, load_golden, oracle_forward, strict_share
from oracle import blocks as ob
from oracle import nms as onms


def test_kat_maxpool21(golden_dir):
    g = load_golden("kat")
    got = ob.max_pool(torch.arange(16.).view(1, 1, 4, 4), 2, 1).numpy()
    assert np.array_equal(got, g["maxpool21"])
    assert np.array_equal(got[0, 0], np.array([[5, 6, 7, 6], [9, 10, 11, 10], [13, 14, 15, 14], [9, 10, 11, 10]], np.float32))


def test_kat_nms():
    g = load_golden("kat")
    pred = C.NMS_KAT_ROWS.copy()
    dets, kept = onms.nms_image(pred, **C.NMS_KAT_ARGS)
    assert np.allclose(dets, C.NMS_KAT_EXPECT, atol=1e-4)
    assert np.allclose(dets, g["nms_kat"], rtol=0, atol=1e-5)
    assert np.array_equal(dets[:, 4:], g["nms_kat"][:, 4:])
    assert np.array_equal(pred[:, 4], g["nms_kat_col4"])           # in-place conf mutation (utils.py:213)
    assert np.allclose(pred[:, 4], C.NMS_KAT_COL4, atol=1e-6)
    assert kept.tolist() == [0, 2]


def test_kat_downsample_sub_is_pre_add():
    from oracle.models import darknet_stage
    from pytorch_yolo_amd.models.yolov3_spp import DownSample
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    g = load_golden("kat")
    sd = synth_state_dict(DownSample(4, 8, repeat=1).state_dict(), 5)
    sd = {"s." + k: v for k, v in sd.items()}
    x, sub = darknet_stage(sd, "s", synth_images(1, 16, 16, 6, channels=4), 2)
    assert np.array_equal(x.numpy(), g["downsample_x"]) and np.array_equal(sub.numpy(), g["downsample_sub"])
    assert not np.array_equal(x.numpy(), sub.numpy())


@pytest.mark.parametrize("name", list(C.MODEL_CASES))
def test_model_small_bit_exact(name):
    case = C.MODEL_CASES[name]
    model, sd, x = build_case(case)
    io, p = oracle_forward(case, sd, x)
    g = load_golden("model_" + name)
    assert np.array_equal(io.numpy(), g["io"])
    for k, t in enumerate(p):
        assert np.array_equal(t.numpy(), g[f"p{k}"])


@pytest.mark.parametrize("name", ["tiny_small", "spp_small"])
def test_model_fused_matches(name):
    """fold_bn restatement: oracle forward on product-fused weights vs reference fused forward."""
    case = C.MODEL_CASES[name]
    model, sd, x = build_case(case)
    model.fuse()
    io, _ = oracle_forward(case, model.state_dict(), x)
    g = load_golden("model_" + name)
    np.testing.assert_allclose(io.numpy(), g["io_fused"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(io.numpy(), g["io"], rtol=1e-3, atol=1e-3)      # fuse() vs un-fused (6e-5)


@pytest.mark.parametrize("name", list(C.FULL_CASES))
def test_model_full_size_samples(name):
    case = C.FULL_CASES[name]
    model, sd, x = build_case(case)
    io, p = oracle_forward(case, sd, x)
    g = load_golden("full_" + name)
    assert list(io.shape) == g["io_shape"].tolist()
    assert np.array_equal(io.numpy()[:, g["rows"]], g["io_rows"])
    np.testing.assert_allclose(io.numpy().astype(np.float64).sum(1), g["io_colsum"], rtol=1e-9)
    for k, t in enumerate(p):
        assert list(t.shape) == g[f"p{k}_shape"].tolist()
    dets, kept = onms.non_max_suppression(io.numpy().copy(), **C.NMS_FULL)
    _check_nms(dets, kept, g, prefix="nms_")


def test_separable_full_size_case():
    """tests/_cases.py::SEPARABLE (round 3): the oracle reproduces the reference on the conditioned full-size case - sampled rows
    and column sums of io bit / 1e-9 equal, the 22 detections with identical kept indices - the calibration that the golden stores
    is what `calibrate_separable_heads` derives from the oracle's own raw heads (so the stored arrays are data, not a free choice),
    and the oracle's bf16-policy re-run pairs with it strictly in both directions (the property the seed was selected for)."""
    from oracle import models as om
    from oracle.policy import run_policy
    from pytorch_yolo_amd.utils.synthetic import synth_state_dict
    sep = C.SEPARABLE
    model, sd, x, g = build_separable_case()
    with torch.no_grad():
        io, p = om.spp_forward(sd, x, C.SPP_ANCHORS, 80)
    assert list(io.shape) == g["io_shape"].tolist()
    assert np.array_equal(io.numpy()[:, g["rows"]], g["io_rows"])
    np.testing.assert_allclose(io.numpy().astype(np.float64).sum(1), g["io_colsum"], rtol=1e-9)
    dets, kept = onms.non_max_suppression(io.numpy().copy(), sep["conf_thres"], sep["nms_thres"])
    _check_nms(dets, kept, g, prefix="nms_")
    assert int(g["nms_count_0"]) == 22 and float(g["nms_dets_0"][:, 4].min()) > 0.8 and int(g["n_between_04_06"]) == 0
    conf = g["nms_dets_0"][:, 4]
    assert len(np.unique(conf)) == len(conf)                                   # no ties: the reference's argsort order is defined
    # the stored head BN = the calibration rule applied to the un-calibrated model's raw heads
    sd0 = synth_state_dict(model.state_dict(), sep["weight_seed"], n_class=80)
    with torch.no_grad():
        _, p0 = om.spp_forward(sd0, x, C.SPP_ANCHORS, 80)
    wk = [h + ".sequence.batch_norm.weight" for h in C.SEPARABLE_HEADS]
    bk = [h + ".sequence.batch_norm.bias" for h in C.SEPARABLE_HEADS]
    new_w, new_b = C.calibrate_separable_heads([sd0[k].numpy() for k in wk], [sd0[k].numpy() for k in bk], [t[0].numpy() for t in p0], 80,
                                               sep["per_anchor"], sep["span"], sep["gamma_obj"], sep["cls_gain"])
    for k in range(3):
        assert np.array_equal(new_w[k], g[f"head_bn_weight_{k}"]) and np.array_equal(new_b[k], g[f"head_bn_bias_{k}"])
    # the rounding model (fp32 oracle re-run under the product's bf16 rounding points) carries these detections strictly
    io_b, _ = run_policy(om.spp_forward, sd, x, C.SPP_ANCHORS, 80, policy="bf16")
    db, _ = onms.non_max_suppression(io_b.numpy().copy(), sep["conf_thres"], sep["nms_thres"])
    assert strict_share(g["nms_dets_0"], db[0]) == 1.0 and strict_share(db[0], g["nms_dets_0"]) == 1.0


def test_rule_selected_cases_are_what_the_rule_selects():
    """tests/_cases.py::RULE_SEEDS: (1) scanning patch-image seeds upward from 1, the committed seeds are exactly
    the first len(RULE_SEEDS) that satisfy ``rule_verdict`` on the fp32 path's own outputs - no seed was skipped, none chosen by
    what a bf16 run makes of it; (2) per committed case the oracle reproduces the reference golden (sampled io rows bit-equal, NMS
    kept set / conf / class bit-equal), the stored head BN is ``calibrate_rule_heads`` applied to the un-calibrated raw heads, every
    cut gap is >= 8 drifts, no conf lies within 0.05 of the threshold and the detections are tie-free."""
    from oracle import models as om
    from pytorch_yolo_amd import YOLOv3SPP
    from pytorch_yolo_amd.utils.synthetic import synth_state_dict
    rule = C.RULE
    sd0 = synth_state_dict(YOLOv3SPP(anchors=C.SPP_ANCHORS).state_dict(), rule["weight_seed"], n_class=80)
    qualifying = []
    seed = 0
    while len(qualifying) < len(C.RULE_SEEDS):
        seed += 1
        assert seed <= max(C.RULE_SEEDS), "a committed seed does not qualify, or an earlier qualifying seed was skipped"
        x = torch.from_numpy(C.patch_image(seed, rule["n_patches"]))
        with torch.no_grad():
            _, p0 = om.spp_forward(sd0, x, C.SPP_ANCHORS, 80)
            sd, gaps = C.rule_state_dict(sd0, [t[0].numpy() for t in p0], rule)
            io, _ = om.spp_forward(sd, x, C.SPP_ANCHORS, 80)
        ok, why = C.rule_verdict(io.numpy()[0], gaps, rule)
        if not ok:
            continue
        qualifying.append(seed)
        g = load_golden(f"full_spp_640_rule_{seed}")
        assert np.array_equal(io.numpy()[:, g["rows"]], g["io_rows"]), "oracle differs from the reference on a rule case"
        for k, h in enumerate(C.SEPARABLE_HEADS):
            assert np.array_equal(sd[h + ".sequence.batch_norm.weight"].numpy(), g[f"head_bn_weight_{k}"])
            assert np.array_equal(sd[h + ".sequence.batch_norm.bias"].numpy(), g[f"head_bn_bias_{k}"])
        live = [v for v in gaps if v is not None]
        assert min(live) >= rule["min_gap_drifts"] * rule["drift"] and np.allclose([-1.0 if v is None else v for v in gaps], g["cut_gaps"])
        dets, kept = onms.non_max_suppression(io.numpy().copy(), rule["conf_thres"], rule["nms_thres"])
        _check_nms(dets, kept, g, "nms_")
        conf = g["nms_dets_0"][:, 4]
        assert len(conf) >= rule["min_detections"] and len(np.unique(conf)) == len(conf) and conf.min() > rule["conf_thres"] + rule["conf_band"]
    assert tuple(qualifying) == tuple(C.RULE_SEEDS)


def _check_nms(dets, kept, g, prefix=""):
    for b, (d, k) in enumerate(zip(dets, kept)):
        n = int(g[f"{prefix}count_{b}"])
        if n == 0:
            assert d is None
            continue
        ref = g[f"{prefix}dets_{b}"]
        assert d.shape == ref.shape
        assert np.array_equal(k, g[f"{prefix}kept_{b}"]), "kept-index set differs from the reference"
        assert np.array_equal(d[:, 4:], ref[:, 4:]), "conf / class_conf / class must be bit-equal"
        # merged boxes: torch's reduction order vs sequential fp32 -> a few ulp
        np.testing.assert_allclose(d[:, :4], ref[:, :4], rtol=2e-6, atol=2e-4)


@pytest.mark.parametrize("name", list(C.NMS_CASES))
def test_nms_cases(name):
    pred, conf, iou = C.nms_case_inputs(name)
    g = load_golden(name)
    work = pred.copy()
    dets, kept = onms.non_max_suppression(work, conf, iou)
    _check_nms(dets, kept, g)
    for b in range(pred.shape[0]):
        assert np.array_equal(work[b, :, 4], g[f"col4_{b}"], equal_nan=True)
    if name == "nms_none_pass":
        assert dets[1] is None and dets[0] is not None
    if name == "nms_dense_nc3":
        assert max(len(d) for d in dets) <= 3 * 100


def test_scale_coords_kat():
    g = load_golden("kat")
    boxes = C.scale_coords_boxes()
    for i, (s1, s0) in enumerate(C.SCALE_CASES):
        assert np.array_equal(onms.scale_coords(s1, boxes, s0), g[f"scale_{i}"]), (s1, s0)
