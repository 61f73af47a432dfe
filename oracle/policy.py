"""Precision-policy emulation on top of the oracle.  TEST INFRASTRUCTURE (tests/, tests/diag/ only).

The oracle (oracle/models.py) is the reference's fp32 arithmetic.  The product's fast path keeps the same graph but
rounds at documented points: the BN is folded into the weights in fp32, conv operands - activations and
folded weights - are bf16, products are accumulated in fp32, a residual sum is formed in fp32 and stored as bf16 once per
unit, detection-head convs output fp32.  ``run_policy`` re-runs an oracle forward under exactly those rounding points, so
that

  * tests can hold the HIP path to its OWN specification much tighter than to the fp32 reference (what is left is the
    fp32 summation order and bf16 double-rounding flips), and
  * the distance between the two CPU runs (fp32 vs bf16 policy) is the error budget of the bf16 path, layer by layer
    (tests/diag/drift_model.py -> profiles/r02_drift_model.md), with no GPU in the loop.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import blocks as ob


def bf16r(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


def fold_state_dict(sd):
    """Every ConvBlock's BN folded into a biased conv in fp32 (reference utils/torch_utils.py:33-60 via oracle.blocks.fold_bn),
    keyed like the reference's fuse() output (``<block>.sequence.0.{weight,bias}``); plain convs are passed through."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".sequence.conv.weight"):
            p = k[:-len(".sequence.conv.weight")]
            bn = p + ".sequence.batch_norm."
            w, b = ob.fold_bn(v, sd[bn + "weight"], sd[bn + "bias"], sd[bn + "running_mean"], sd[bn + "running_var"])
            out[p + ".sequence.0.weight"], out[p + ".sequence.0.bias"] = w, b
        elif ".sequence.batch_norm." not in k:
            out[k] = v
    return out


def run_policy(forward, sd, x, *args, policy: str = "bf16", taps=None, select=None):
    """``forward(sd, x, *args)`` (an oracle.models forward) under a precision policy:
    'fp32' (the oracle itself), 'bf16' (the product's rounding points), 'bf16_f32stream' (bf16 operands, residual stream
    kept in fp32).  ``taps``: optional dict filled with every intermediate tensor by block name.
    ``select(name) -> bool`` (bf16 policies): only the rounding points it accepts are applied - ``name`` is a conv's block prefix
    ("down3.seq2.1", "branch1_2.conv2", ...) for its operand rounding and "<stage>.add<i>" for the stream store; everything else
    stays fp32 (BN folded all the same).  tests/diag/drift_attribution.py uses it to split the head-logit drift by stage."""
    if policy not in ("fp32", "bf16", "bf16_f32stream"):
        raise ValueError(policy)
    real_conv = F.conv2d
    run_sd = sd if policy == "fp32" else fold_state_dict(sd)
    owner = {v.data_ptr(): k[:-len(".sequence.0.weight")] for k, v in run_sd.items() if k.endswith(".sequence.0.weight")} if select else {}

    def conv(xx, w, b=None, **kw):
        if policy == "fp32" or (select is not None and not select(owner.get(w.data_ptr(), "?"))):
            return real_conv(xx, w, b, **kw)
        return real_conv(bf16r(xx), bf16r(w), b, **kw)

    def tap(name, t):
        r = bf16r(t) if (policy == "bf16" and ".add" in name and (select is None or select(name))) else None     # the stream is stored as bf16, once per unit
        if taps is not None:
            taps[name] = t if r is None else r
        return r

    prev = ob.set_tap(tap)
    ob.F.conv2d = conv
    try:
        with torch.no_grad():
            return forward(run_sd, x, *args)
    finally:
        ob.F.conv2d = real_conv
        ob.set_tap(prev)
