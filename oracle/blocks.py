"""Oracle building blocks (functional, fp32, CPU torch).  TEST INFRASTRUCTURE.

Each function names the reference lines it restates.  Weights come from a
``state_dict`` with the reference's key names; ``prefix`` selects one block.
Both the un-fused naming (``<prefix>.sequence.conv.weight`` +
``.sequence.batch_norm.*``) and the post-``fuse()`` naming
(``<prefix>.sequence.0.{weight,bias}``) are accepted.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm2d default, /root/reference/pytorch_yolo/models/yolo_base.py:37
LEAKY_SLOPE = 0.1      # yolo_base.py:38

# Optional observer of intermediate tensors (tools/drift_trace.py, tests): called as tap(name, tensor) with the block's
# state_dict prefix ("down3.seq2.1", ...; residual sums as "<stage>.add<i>").  None: no overhead, nothing changes.
# An observer may return a tensor to take the place of the observed one (precision-policy emulation in
# tools/drift_model.py: e.g. round the residual stream to bf16); returning None leaves the value alone.
_TAP = None


def set_tap(fn):
    """Install (or with None remove) the observer; returns the previous one."""
    global _TAP
    prev, _TAP = _TAP, fn
    return prev


def tap(name, t):
    if _TAP is not None:
        r = _TAP(name, t)
        if r is not None:
            return r
    return t


def conv_bn_leaky(sd, prefix, x, stride=1, pad=True):
    """ConvBlock.forward — yolo_base.py:19-44.

    y = leaky_0.1( BN_eval( conv(x, W, stride, pad=(k-1)//2, no bias) ) )
    After ConvBlock.fuse() (yolo_base.py:46-57) the block is a biased conv +
    leaky; both spellings give the same function up to fp32 rounding.
    """
    fused_w = sd.get(prefix + ".sequence.0.weight")
    if fused_w is not None:
        k = fused_w.shape[-1]
        y = F.conv2d(x, fused_w, sd[prefix + ".sequence.0.bias"], stride=stride,
                     padding=(k - 1) // 2 if pad else 0)
        return tap(prefix, F.leaky_relu(y, LEAKY_SLOPE))
    w = sd[prefix + ".sequence.conv.weight"]
    k = w.shape[-1]
    y = F.conv2d(x, w, None, stride=stride, padding=(k - 1) // 2 if pad else 0)
    bn = prefix + ".sequence.batch_norm."
    y = F.batch_norm(y, sd[bn + "running_mean"], sd[bn + "running_var"],
                     sd[bn + "weight"], sd[bn + "bias"], training=False, eps=BN_EPS)
    return tap(prefix, F.leaky_relu(y, LEAKY_SLOPE))


def plain_conv1x1(sd, prefix, x):
    """nn.Conv2d(C, 255, kernel_size=1) with bias — yolov3_tiny.py:38,42."""
    return tap(prefix, F.conv2d(x, sd[prefix + ".weight"], sd[prefix + ".bias"]))


def fold_bn(conv_w, gamma, beta, mean, var, conv_b=None, eps=BN_EPS):
    """fuse_conv_and_bn — /root/reference/pytorch_yolo/utils/torch_utils.py:33-60.

    W' = diag(gamma / sqrt(eps + var)) @ W ;  b' = b + beta - gamma*mean/sqrt(var+eps)
    (the reference forms the diag matrix and uses torch.mm, :48-50).
    """
    cout = conv_w.shape[0]
    scale = gamma / torch.sqrt(eps + var)
    w = torch.mm(torch.diag(scale), conv_w.reshape(cout, -1)).reshape(conv_w.shape)
    b0 = conv_b if conv_b is not None else torch.zeros(cout)
    b = b0 + (beta - gamma * mean / torch.sqrt(var + eps))
    return w, b


def max_pool(x, size, stride):
    """MaxPool — yolo_base.py:60-66.  (2,1) is the pad-1 / dilation-2 special."""
    if size == 2 and stride == 1:
        return F.max_pool2d(x, 2, 1, padding=1, dilation=2)
    return F.max_pool2d(x, size, stride, padding=(size - 1) // 2)


def upsample2(x):
    """Upsample(2) nearest — yolo_layer.py:6-13."""
    return F.interpolate(x, scale_factor=2, mode="nearest")


def yolo_decode(p_raw, anchors, n_class, img_size):
    """YOLOLayer.forward (eval, non-onnx) + create_grids — yolo_layer.py:57-69,90-111.

    p_raw: [bs, 3*(5+nc), ny, nx] -> (io [bs, 3*ny*nx, 5+nc], p [bs,3,ny,nx,5+nc]).
    """
    bs, _, ny, nx = p_raw.shape
    na = len(anchors)
    stride = img_size / max(nx, ny)                                   # :102 (python float)
    yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
    grid_xy = torch.stack((xv, yv), 2).float().view(1, 1, ny, nx, 2)  # :105-106
    anchor_vec = torch.tensor(anchors, dtype=torch.float32) / stride  # :109
    anchor_wh = anchor_vec.view(1, na, 1, 1, 2)                       # :110
    p = p_raw.view(bs, na, n_class + 5, ny, nx).permute(0, 1, 3, 4, 2).contiguous()  # :67-69
    io = p.clone()                                                    # :90
    io[..., 0:2] = torch.sigmoid(io[..., 0:2]) + grid_xy              # :91
    io[..., 2:4] = torch.exp(io[..., 2:4]) * anchor_wh                # :92
    io[..., 4:] = torch.sigmoid(io[..., 4:])                          # :93
    io[..., :4] *= stride                                             # :94
    if n_class == 1:
        io[..., 5] = 1                                                # :95-96
    return io.view(bs, -1, 5 + n_class), p                            # :99
