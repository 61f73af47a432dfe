import os, sys, torch
sys.path.insert(0, '/root/repo')
from pytorch_yolo_amd import kernels as K
buf = torch.randn(32, 20, 20, 2048, device='cuda').to(torch.bfloat16)
for _ in range(5): K.spp(buf, n=32, h=20, w=20, c=512)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): K.spp(buf, n=32, h=20, w=20, c=512)
e1.record(); torch.cuda.synchronize()
print("spp ms", e0.elapsed_time(e1) / 200, "lines" if not os.environ.get("YOLO_SPP_NO_LINES") else "8ch")
