from .lite_yolo import LiteYOLOv3
from .yolov3 import YOLOv3
from .yolov3_spp import YOLOv3SPP
from .yolov3_tiny import YOLOv3Tiny
from .yolov3_tiny_efficient import YOLOv3TinyEfficient
from .yolov3_tiny_mobilenet import YOLOv3TinyMobile
from .yolov3_tiny_shuffle import YOLOv3TinyShuffle
from .yolov3_tiny_squeeze import YOLOv3TinySqueeze

__all__ = ["YOLOv3SPP", "YOLOv3Tiny", "YOLOv3TinyMobile", "YOLOv3TinySqueeze", "YOLOv3TinyShuffle", "YOLOv3TinyEfficient", "YOLOv3", "LiteYOLOv3"]
