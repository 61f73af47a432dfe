"""Why is bench.py's partitioned launch loop 13 % slower after detect_stream() has run?  Times the loop before / after, in one process."""
import gc, importlib.util, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
from pytorch_yolo_amd.utils.utils import nms_capacity
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["spp"]
model = wl["cls"](**wl["kw"]).eval(); model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80)); model = model.to(dev)
x = synth_images(32, 640, 640, 0).to(dev)
plan = model.plan_for(x)
cap = nms_capacity(plan.rows_total, 80)
outs = [(torch.empty((32, cap, 7), device=dev), torch.empty((32, cap), dtype=torch.int32, device=dev), torch.empty((32,), dtype=torch.int32, device=dev)) for _ in range(2)]
order = sys.argv[1] if len(sys.argv) > 1 else "loop,stream,loop,drop,loop,detect,loop"
part = os.environ.get("PART", "1") == "1"
def loop(n=60):
    for i in range(10): plan.launch_detect(x, None, (None,)*3, outs[i % 2], 0.1, 0.5, join=False, whole_batch=True, cu_partition=part, compact=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): plan.launch_detect(x, None, (None,)*3, outs[i % 2], 0.1, 0.5, join=False, whole_batch=True, cu_partition=part, compact=True)
    torch.cuda.synchronize(); return 32 * n / (time.perf_counter() - t)
with torch.no_grad():
    for what in order.split(","):
        if what == "loop": print(f"loop (cu_partition={part}): {loop():.0f} images/s", flush=True)
        elif what == "stream":
            t = time.perf_counter(); n = sum(len(r) for r in model.detect_stream((x for _ in range(60)), 0.1, 0.5)); torch.cuda.synchronize()
            print(f"detect_stream: {n / (time.perf_counter() - t):.0f} images/s", flush=True)
        elif what == "detect":
            t = time.perf_counter()
            for _ in range(20): model.detect(x, 0.1, 0.5)
            torch.cuda.synchronize(); print(f"detect: {32 * 20 / (time.perf_counter() - t):.0f} images/s", flush=True)
        elif what == "drop":
            plan.__dict__.pop("_stream_rings", None); gc.collect(); torch.cuda.empty_cache(); print("dropped the stream rings", flush=True)
