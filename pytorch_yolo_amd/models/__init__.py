from .lite_yolo import LiteYOLOv3
from .yolov3 import YOLOv3
from .yolov3_spp import YOLOv3SPP
from .yolov3_tiny import YOLOv3Tiny
from .yolov3_tiny_mobilenet import YOLOv3TinyMobile

__all__ = ["YOLOv3SPP", "YOLOv3Tiny", "YOLOv3TinyMobile", "YOLOv3", "LiteYOLOv3"]
