import os, sys, torch
sys.path.insert(0, os.getcwd())
from pytorch_yolo_amd import YOLOv3SPP
from pytorch_yolo_amd.utils.synthetic import synth_images
import bench
for reuse in ("1", "0"):
    os.environ["YOLO_REUSE_BUFFERS"] = reuse
    m = YOLOv3SPP(anchors=bench.SPP_ANCHORS).eval().to("cuda:0")
    x = synth_images(32, 640, 640, 0).to("cuda:0")
    plan = m.plan_for(x)
    print("reuse", reuse, "activation GB", plan.activation_bytes() / 1e9, "shared", [p.shared_buffers for p in plan.subs], "algorithmic GB/step", plan.algorithmic_bytes() / 1e9)
