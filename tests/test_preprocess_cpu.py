"""CPU checks of the pre-processing oracle and of the host geometry (no GPU): the letterbox geometry against values
worked out by hand from reference utils/augs.py:24-63, the host mirror against the oracle on a sweep, and the
INTER_AREA restatement against properties that hold for any correct area resampler."""
import numpy as np
import pytest

from oracle import preprocess as O
from pytorch_yolo_amd.utils import augs as A

# (rows, cols, new_shape) -> (target_h, target_w, resize_h, resize_w, pad_top, pad_left, pad_bottom, pad_right)
GEOMETRY_KAT = [
    ((480, 640, 416), (320, 416, 312, 416, 4, 0, 4, 0)),        # r = .75 -> ceil(.75*13) = 10 -> 320; mod(8, 32) / 2 = 4
    ((640, 480, 416), (416, 320, 416, 312, 0, 4, 0, 4)),
    ((1080, 1920, 640), (384, 640, 360, 640, 12, 0, 12, 0)),    # ceil(.5625*20) = 12 -> 384; ratio 1/3
    ((375, 500, (416, 416)), (416, 416, 312, 416, 52, 0, 52, 0)),   # fixed shape: (416-312)/2
    ((32, 32, 416), (416, 416, 416, 416, 0, 0, 0, 0)),          # up-scaling by 13
    ((100, 37, 320), (320, 128, 320, 118, 0, 5, 0, 5)),         # r > 1: ceil(.37*10) = 4 -> 128; round(118.4) = 118
]


@pytest.mark.parametrize("args,want", GEOMETRY_KAT)
def test_letterbox_geometry_kat(args, want):
    for fn in (O.letterbox_params, A.letterbox_params):
        p = fn(*args)
        got = (p["target_height"], p["target_width"], p["resize_height"], p["resize_width"], p["pad_top"], p["pad_left"],
               p["pad_bottom"], p["pad_right"])
        assert got == want
        assert p["resize_ratio"] == max(want[0], want[1]) / max(args[0], args[1])


def test_host_geometry_equals_oracle_on_a_sweep():
    rng = np.random.default_rng(5)
    for _ in range(400):
        h, w = int(rng.integers(8, 2200)), int(rng.integers(8, 2200))
        ns = int(rng.choice([320, 416, 512, 608, 640])) if rng.random() < 0.7 else (int(rng.choice([416, 640])),) * 2
        assert O.letterbox_params(h, w, ns) == A.letterbox_params(h, w, ns)
    shapes = [(320, 416), (416, 416), (384, 640), (416, 320)]
    new_h, new_w, offs = A.equalize_offsets(shapes)
    x, ooffs = O.equalize_shapes([np.zeros((3, h, w), np.float32) for h, w in shapes])
    assert (new_h, new_w) == x.shape[2:] == (416, 640) and offs == ooffs
    assert offs[0] == (48, 112)                 # int(round(48 - .1)), int(round(112 - .1)) (dataset_csv.py:160-161)


def test_area_resize_properties():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (24, 36, 3), dtype=np.uint8)
    # ratio 1: identity
    assert np.array_equal(O.resize_area(img, 1.0, 24, 36), img)
    # constant image stays constant under any ratio (weights of every output sum to 1)
    const = np.full((30, 50, 3), 137, dtype=np.uint8)
    for ratio in (0.37, 0.5, 0.8125, 1.7):
        out = O.resize_area(const, ratio, int(round(30 * ratio)), int(round(50 * ratio)))
        assert np.all(out == 137)
    # exact 3x down-scaling: every output is the mean of a 3x3 block, rounded half to even
    out = O.resize_area(img, 1 / 3, 8, 12)
    blocks = img.reshape(8, 3, 12, 3, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.max(np.abs(out.astype(np.float64) - blocks)) <= 0.5 + 1e-4
    # the weight tables: every destination's weights sum to 1, indices stay inside the source
    for ssize, dsize, scale in ((640, 416, 640 / 416), (1920, 640, 3.0), (719, 360, 1 / 0.5003909304143862), (37, 118, 1 / 3.2)):
        tab = (O.area_tab if scale >= 1 else O.linear_tab)(ssize, dsize, scale)
        assert len(tab) == dsize
        for ent in tab:
            assert abs(sum(float(w) for _, w in ent) - 1.0) < 1e-5 and all(0 <= s < ssize for s, _ in ent)


def test_letterbox_and_batch_oracle():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (60, 100, 3), dtype=np.uint8)
    out, p = O.letterbox(img, 64)
    assert out.shape == (p["target_height"], p["target_width"], 3) == (64, 64, 3) or out.shape[1] == 64
    top, left = p["pad_top"], p["pad_left"]
    inner = out[top:top + p["resize_height"], left:left + p["resize_width"]]
    assert np.array_equal(out[0], out[top]) and np.array_equal(out[-1], inner[-1])          # replicate border
    x, metas = O.preprocess_batch([img, rng.integers(0, 256, (100, 60, 3), dtype=np.uint8)], 64)
    assert x.dtype == np.float32 and x.shape[0] == 2 and 0.0 <= x.min() and x.max() <= 1.0
    m = metas[0]
    sub = x[0, :, m["off_y"]:m["off_y"] + m["target_height"], m["off_x"]:m["off_x"] + m["target_width"]]
    assert np.array_equal(sub, O.convert_img_for_net(out))
    if m["off_y"] or m["off_x"]:
        assert x[0, 0, 0, 0] == np.float32(0.5)
