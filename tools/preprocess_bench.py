"""Throughput of the device pre-processing (LetterBox + /255 + CHW + equalize) and of pre-processing + detect()."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytorch_yolo_amd import YOLOv3SPP
from pytorch_yolo_amd.utils.augs import preprocess_batch
from pytorch_yolo_amd.utils.synthetic import synth_state_dict

SPP_ANCHORS = [[(10, 13), (16, 30), (33, 23)], [(30, 61), (62, 45), (59, 119)], [(116, 90), (156, 198), (373, 326)]]


def main():
    dev = torch.device("cuda", 0)
    bs, h, w = 32, 1080, 1920
    g = torch.Generator(device="cpu").manual_seed(0)
    imgs = [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, generator=g).to(dev) for _ in range(bs)]
    x, meta = preprocess_batch(imgs, 640)
    out = torch.empty_like(x)
    for _ in range(3):
        preprocess_batch(imgs, 640, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        preprocess_batch(imgs, 640, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"preprocess {bs} x {h}x{w}x3 u8 -> {tuple(x.shape)} f32: {dt * 1e3:.3f} ms  ({bs / dt:.0f} images/s, "
          f"{(bs * h * w * 3 + x.numel() * 4) / dt / 1e9:.0f} GB/s algorithmic)", flush=True)
    model = YOLOv3SPP(anchors=SPP_ANCHORS).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
    model = model.to(dev)
    with torch.no_grad():
        for _ in range(3):
            model.detect(preprocess_batch(imgs, 640, out=out)[0], 0.1, 0.5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            dets = model.detect(preprocess_batch(imgs, 640, out=out)[0], 0.1, 0.5)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"preprocess + detect() on {tuple(x.shape)} (host-synchronous API, list output): {dt * 1e3:.3f} ms  ({bs / dt:.0f} images/s)")


if __name__ == "__main__":
    main()
