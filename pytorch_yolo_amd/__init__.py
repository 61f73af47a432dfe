"""MI355X-native YOLOv3 inference hot path, drop-in behind the Dipet/pytorch_yolo model API.

    from pytorch_yolo_amd import YOLOv3SPP, non_max_suppression
    model = YOLOv3SPP(anchors=...).eval().cuda()
    io, p = model(x)                       # same return structure as the reference forward
    dets = non_max_suppression(io, 0.1, 0.5)

Everything numeric runs in csrc/libyolo_hip.so (hand-written HIP for gfx950); there is no
CPU or eager fallback.
"""
import os as _os

# HIP multiplexes its streams over GPU_MAX_HW_QUEUES hardware queues (4 by default).  One detect() pipeline per sub-batch, the
# streams of the joined calls, the all-gather's side stream and RCCL's own make 6-7: on 4 queues two of them share one, and two
# sub-batch pipelines on one queue run one after the other (measured: 5,740 -> 4,100 images/s in the sharded bench, -3 % in the
# one-GPU one).  Read by the HIP runtime when it initialises, i.e. at the first torch.cuda call: import this package before that,
# or export the variable yourself.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    import sys as _sys
    _torch = _sys.modules.get("torch")
    if _torch is not None and _torch.cuda.is_initialized():      # too late: the runtime has read its default of 4
        import warnings as _warnings
        _warnings.warn("pytorch_yolo_amd was imported after torch.cuda was initialised and GPU_MAX_HW_QUEUES is not set: HIP keeps 4 "
                       "hardware queues and the detect() pipelines will share them (-3 % on one GPU, -29 % when sharded). Export "
                       "GPU_MAX_HW_QUEUES=8 or import pytorch_yolo_amd before the first torch.cuda call.", RuntimeWarning, stacklevel=2)

from .models import (LiteYOLOv3, YOLOv3, YOLOv3SPP, YOLOv3Tiny, YOLOv3TinyEfficient, YOLOv3TinyMobile, YOLOv3TinyShuffle,
                     YOLOv3TinySqueeze)
from .utils.utils import non_max_suppression

__all__ = ["YOLOv3SPP", "YOLOv3Tiny", "YOLOv3TinyMobile", "YOLOv3TinySqueeze", "YOLOv3TinyShuffle", "YOLOv3TinyEfficient", "YOLOv3", "LiteYOLOv3", "non_max_suppression"]
