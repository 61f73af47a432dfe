"""MI355X-native YOLOv3 inference hot path, drop-in behind the Dipet/pytorch_yolo model API.

    from pytorch_yolo_amd import YOLOv3SPP, non_max_suppression
    model = YOLOv3SPP(anchors=...).eval().cuda()
    io, p = model(x)                       # same return structure as the reference forward
    dets = non_max_suppression(io, 0.1, 0.5)

Everything numeric runs in csrc/libyolo_hip.so (hand-written HIP for gfx950); there is no
CPU or eager fallback.
"""
from .models import (LiteYOLOv3, YOLOv3, YOLOv3SPP, YOLOv3Tiny, YOLOv3TinyEfficient, YOLOv3TinyMobile, YOLOv3TinyShuffle,
                     YOLOv3TinySqueeze)
from .utils.utils import non_max_suppression

__all__ = ["YOLOv3SPP", "YOLOv3Tiny", "YOLOv3TinyMobile", "YOLOv3TinySqueeze", "YOLOv3TinyShuffle", "YOLOv3TinyEfficient", "YOLOv3", "LiteYOLOv3", "non_max_suppression"]
