"""Deterministic synthetic weights / inputs (pure numpy RNG, no torch RNG).

SURVEY.md §8(d): the bench, the parity tests and the golden-vector script all
draw from this generator, so the GPU box, this container and the committed
fixtures see the same tensors without shipping any weights.

Every tensor is seeded from (seed, crc32(key)) so the values do not depend on
the order the keys are visited in, and a sub-model gets the same weights for
the keys it shares with a bigger one.
"""
from __future__ import annotations

import re
import zlib
from collections import OrderedDict

import numpy as np
import torch

__all__ = ["synth_state_dict", "synth_images", "key_rng", "calibrate_plain_heads"]


def key_rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(key.encode("utf-8"))])


def _conv_weight(rng, shape, gain=1.0):
    fan_in = int(np.prod(shape[1:]))
    std = gain * np.sqrt(2.0 / fan_in)
    return (rng.standard_normal(shape) * std).astype(np.float32)


def _head_channel_stats(n_ch: int, n_class: int, bn_leaky: bool):
    """Per-output-channel (scale, shift) of a detection head's logits, by the
    role of channel c = a*(5+nc)+k (yolo_layer.py:67-69), so that decode/NMS see
    a realistic spread (a few % of rows above conf 0.1) instead of the all-0.5
    scores of default init (SURVEY.md §7 "Hard parts").

    Plain-conv heads (tiny): logit ~ N(shift, scale).  ConvBlock heads
    (YOLOv3-SPP: BN + LeakyReLU(0.1) after the head conv, yolov3_spp.py:86,99,111):
    the values are BN gamma/beta; the negative side is x10 because the leaky
    slope divides it by 10 again before the sigmoid.
    """
    k = np.arange(n_ch) % (5 + n_class)
    pick = lambda xy, wh, obj, cls: np.where(k < 2, xy, np.where(k < 4, wh, np.where(k == 4, obj, cls)))
    scale, shift = HEAD_STATS["bn_leaky" if bn_leaky else "plain"]
    scale, shift = pick(*scale), pick(*shift)
    return scale.astype(np.float32), shift.astype(np.float32)


# (xy, wh, obj, cls) logit scale and shift per head flavour; tuned so that SPP-640 /
# tiny-416 on synth_images() leave O(10^2..10^3) rows per image above conf 0.1.
HEAD_STATS = {
    "bn_leaky": ((1.5, 0.3, 25.0, 10.0), (0.0, 0.0, -40.0, -30.0)),   # ~2.4k of 25,200 rows
    "plain": ((1.0, 0.3, 2.0, 1.2), (0.0, 0.0, -3.0, -3.0)),          # ~400 of 2,535 rows
}

# measured pre-BN variance of the SPP head convs under this generator (1234, 640x640)
_SPP_HEAD_VAR = {"branch1_2.conv2.": 50.0, "branch2_3.conv7.": 25.0, "branch3_2.conv7.": 6.0,
                 "seqy3_2.conv7.": 6.0, "seqy3_2.conv2.": 6.0}


def synth_state_dict(template, seed: int = 1234, n_class: int | None = None):
    """Fill a ``state_dict``-shaped mapping with seeded values.

    ``template`` maps key -> tensor (only shapes / dtypes are read).  Rules by
    key suffix (reference key names):

    * ``...conv.weight`` / ``....0.weight`` / 4-D weights: He-normal
    * ``batch_norm.weight`` ~ 0.917*U(0.5, 1.5) (x0.25 on residual tails), ``.bias`` ~ N(0, 0.1),
      ``.running_mean`` ~ N(0, 0.1), ``.running_var`` ~ U(0.5, 1.5)
    * 1-D conv biases ~ N(0, 0.1)
    * with ``n_class`` given, the detection heads get role-dependent logit
      statistics (``_head_channel_stats``): plain-conv heads (tiny) through the
      weight gain and bias; ConvBlock heads (YOLOv3-SPP, BN + LeakyReLU(0.1)
      after the head conv, yolov3_spp.py:86,99,111) through BN gamma/beta, x10
      on the negative side to undo the leaky slope.
    """
    out = OrderedDict()
    # LiteYOLOv3's third head is seqy3_2.conv2; in YOLOv3 the same name is an ordinary mid layer (its head is conv7)
    lite_head = not any("seqy3_2.conv7." in k for k in template)
    _is_head = lambda k: _is_head_key(k) and (lite_head or "seqy3_2.conv2." not in k)   # noqa: E731
    for key, ref in template.items():
        shape = tuple(ref.shape)
        rng = key_rng(seed, key)
        if key.endswith("num_batches_tracked"):
            val = np.zeros(shape, dtype=np.int64)
        elif len(shape) == 4:
            val = _conv_weight(rng, shape)
            if n_class is not None and _is_head(key) and ".sequence." not in key:
                sc, _ = _head_channel_stats(shape[0], n_class, False)   # plain conv head
                val = (val * (sc / np.sqrt(2.0))[:, None, None, None]).astype(np.float32)
        elif key.endswith("running_var"):
            val = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            if n_class is not None and _is_head(key):
                v = next(v for m, v in _SPP_HEAD_VAR.items() if m in key)
                val = (rng.uniform(0.8, 1.2, shape) * v).astype(np.float32)
        elif key.endswith("running_mean"):
            val = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        elif "batch_norm" in key and key.endswith(".weight"):
            # 0.917 = 1/sqrt(E[gamma^2] E[1/var]) keeps the activation scale
            # flat through ~75 eval-mode BN layers; residual-closing BNs
            # (downN.seqM.1) get a further 0.25 so 23 stacked adds do not
            # double the variance each time (no batch statistics at eval).
            val = (rng.uniform(0.5, 1.5, shape) * 0.917).astype(np.float32)
            if _RESIDUAL_TAIL.search(key):
                val = (val * 0.25).astype(np.float32)
            if n_class is not None and _is_head(key):
                sc, _ = _head_channel_stats(shape[0], n_class, True)
                val = (rng.uniform(0.9, 1.1, shape) * sc).astype(np.float32)
        elif "batch_norm" in key and key.endswith(".bias"):
            val = (rng.standard_normal(shape) * 0.1).astype(np.float32)
            if n_class is not None and _is_head(key):
                _, sh = _head_channel_stats(shape[0], n_class, True)
                val = (val + sh).astype(np.float32)
        elif key.endswith(".bias"):
            val = (rng.standard_normal(shape) * 0.1).astype(np.float32)
            if n_class is not None and _is_head(key):
                _, sh = _head_channel_stats(shape[0], n_class, False)
                val = (val + sh).astype(np.float32)
        elif key.endswith(".weight"):  # 1-D scale of a BN without "batch_norm" in its name (MobileNetV2 keys)
            val = (rng.uniform(0.5, 1.5, shape) * 0.917).astype(np.float32)
            if _MBV2_PROJECT_BN.search(key) or key.endswith("._bn2.weight"):   # linear-bottleneck BN feeding an identity add
                val = (val * 0.5).astype(np.float32)                             # (MobileNetV2 / EfficientNet-B0 project BN)
            elif key.endswith("._bn0.weight") or key.endswith("._bn1.weight") or ".stem.1.weight" in key:
                val = (val * 1.6).astype(np.float32)   # swish keeps ~0.36 of a unit normal's power, squeeze-excite ~0.3 more
        else:
            val = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        out[key] = torch.from_numpy(val)
    return out


_MBV2_PROJECT_BN = re.compile(r"features\.sequence\d+\.\d+\.conv\.[23]\.weight$")
_RESIDUAL_TAIL = re.compile(r"down\d+\.seq\d+\.1\.")

_HEAD_MARKERS = (
    "branch1_2.conv2.", "branch2_3.conv7.", "branch3_2.conv7.",   # YOLOv3-SPP heads
    "branch1_conv3.", "branch2_conv2.",                            # tiny / MobileNet heads
    "seq_y1.conv2.", "seqy2_3.conv7.", "seqy3_2.conv7.",           # YOLOv3 heads (plain, plain, ConvBlock)
    "seqy3_2.conv2.",                                              # LiteYOLOv3 head 3 (ConvBlock); heads 1/2 as YOLOv3
)


def _is_head_key(key: str) -> bool:
    return any(m in key for m in _HEAD_MARKERS)


def synth_images(bs: int, h: int, w: int, seed: int = 0, channels: int = 3) -> torch.Tensor:
    """float32 NCHW in [0, 1) — the reference's ``/255`` input contract
    (/root/reference/pytorch_yolo/utils/dataset_csv.py:79-87).

    Multi-scale block noise (64/16/4/1-pixel blocks, nearest-replicated) rather
    than white noise: white noise averages out after a few conv layers, every
    grid cell then sees the same features and all rows of a head score alike;
    blocks give the heads spatially varying logits, i.e. a realistic NMS load.
    """
    rng = np.random.default_rng(int(seed))
    img = np.zeros((bs, channels, h, w), dtype=np.float32)
    for block, amp in ((64, 0.35), (16, 0.3), (4, 0.2), (1, 0.15)):
        gh, gw = -(-h // block), -(-w // block)
        g = rng.random((bs, channels, gh, gw), dtype=np.float32)
        g = np.repeat(np.repeat(g, block, axis=2), block, axis=3)[:, :, :h, :w]
        img += np.float32(amp) * g
    return torch.from_numpy(np.minimum(img, np.float32(0.999999)))


def calibrate_plain_heads(model, x, obj_std: float = 2.0) -> float:
    """Synthetic weights only: rescale the plain (biased 1x1 conv) detection heads so that the objectness logits have the
    spread HEAD_STATS["plain"] intends (std ``obj_std`` around the bias).  The head gains of synth_state_dict assume unit
    variance at the head input; an encoder that shrinks its activations (MobileNetV2's linear bottlenecks under this
    generator) leaves every logit at its bias, sigmoid(-3)^2 < conf_thres, and the NMS leg of a benchmark would be empty.
    Runs one forward on (a few images of) ``x``; returns the factor applied."""
    with torch.no_grad():
        _, p = model(x[: min(4, x.shape[0])])
    obj = torch.cat([(t[..., 4] - t[..., 4].mean()).flatten() for t in p])
    f = float(obj_std / max(float(obj.std()), 1e-6))
    sd = model.state_dict()
    for k, v in sd.items():
        if _is_head_key(k) and v.dim() == 4 and ".sequence." not in k:
            sd[k] = v * f
    model.load_state_dict(sd)
    return f
