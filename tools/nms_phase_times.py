import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
from pytorch_yolo_amd.utils.utils import nms_capacity, nms_launch
wl = bench.WORKLOADS["spp"]; dev = torch.device("cuda", 0)
model = wl["cls"](**wl["kw"]).eval(); model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80)); model = model.to(dev); model.n_streams = 1
x = synth_images(16, 640, 640, 0).to(dev)
with torch.no_grad(): io, _ = model(x)
cap = nms_capacity(io.shape[1], 80)
out = (torch.empty((16, cap, 7), device=dev), torch.empty((16, cap), dtype=torch.int32, device=dev), torch.empty((16,), dtype=torch.int32, device=dev))
for _ in range(3): nms_launch(io, 0.1, 0.5, out, slot=0)
torch.cuda.synchronize()
d = out[0][:, cap - 1].cpu()
print("load, sort, classes, gather, sort2, write, nlist:")
print(d.mean(0).tolist()); print(d[:4].tolist())
