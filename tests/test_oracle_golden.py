"""CPU: the oracle (oracle/) against the golden vectors captured from the reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import _cases as C
from helpers import build_case, load_golden, oracle_forward
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
