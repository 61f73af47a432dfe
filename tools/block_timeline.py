#!/usr/bin/env python3
"""Where does a conv launch spend its time?  Diagnostic build with per-workgroup stamps.

    python tools/block_timeline.py --build | --build-all    # here (CPU): tools/_stamps/libyolo_hip_stamps*.so, -DYOLO_STAMPS
    python tools/block_timeline.py [--variant V] n,h,w,cin,cout,k,stride[,res] ...      # on the GPU box
    python tools/block_timeline.py --model [streams]        # timeline of one forward's launch list (16 images per stream)

Every workgroup of the conv kernels records (s_memrealtime start, end, s_memtime cycles, HW_ID/XCC_ID).  Printed per
layer: launch duration by HIP events, first start -> last end, workgroup duration min/median/max, the in-kernel shader
clock (cycles / realtime), how many workgroups each CU ran, and the idle share of the CU-slots.  The shipped
library has no stamps (the macros expand to nothing).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_stamps")
LIB = os.path.join(OUT, "libyolo_hip_stamps.so")


def build(defines=(), tag=""):
    """-DYOLO_STAMPS plus optional timing-only ablation defines (YOLO_ABL_NOBARRIER, YOLO_ABL_NOLDS) -> its own library"""
    sys.path.insert(0, ROOT)
    from pytorch_yolo_amd import build as B
    out = os.path.join(OUT, tag or "base")
    os.makedirs(out, exist_ok=True)
    objs = []
    for src, extra in B.SOURCES.items():
        o = os.path.join(out, src.replace(".hip", ".o"))
        subprocess.run([B.HIPCC, *B.COMMON, *extra, "-DYOLO_STAMPS", *[f"-D{d}" for d in defines], "-c",
                        os.path.join(B.CSRC, src), "-o", o], check=True)
        objs.append(o)
    lib = LIB.replace(".so", f"_{tag}.so") if tag else LIB
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    print(lib)


def run(spec, reps=5):
    import numpy as np
    import torch
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import ACT_LEAKY01
    vals = [int(v) for v in spec.split(",")]
    n, h, w, cin, cout, k, stride = vals[:7]
    use_res = len(vals) > 7 and vals[7]
    dev = "cuda:0"
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
    wt = torch.randn(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5
    wp, bp, kpad, cout_pad = K.pack_conv_weight(wt, torch.zeros(cout), cin)
    wp, bp = wp.to(dev), bp.to(dev)
    y = torch.empty(n, ho, wo, cout, dtype=torch.bfloat16, device=dev)
    res = torch.randn(n, ho, wo, cout, device=dev).to(torch.bfloat16) if use_res else None
    d = K.conv_desc(n=n, h=h, w=w, cin=cin, in_c_total=cin, in_c_offset=0, cout=cout, out_c_total=cout, out_c_offset=0,
                    ksize=k, stride=stride, act=ACT_LEAKY01, kpad=kpad, cout_pad=cout_pad,
                    res=(cout, 0) if use_res else (0, 0))
    stamps = torch.zeros(1 << 16, 4, dtype=torch.int64, device=dev)
    os.environ["YOLO_STAMP_PTR"] = hex(stamps.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(reps):           # back to back; the stamps of the last launch survive
        e0.record()
        K.conv2d(x, wp, bp, y, d, residual=res)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    s = stamps.cpu().numpy()
    s = s[s[:, 1] > 0]
    t0, t1, cyc, hw = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    span = (t1.max() - t0.min()) * 0.01                    # us (100 MHz)
    dur = (t1 - t0) * 0.01
    clk = cyc.sum() / ((t1 - t0).sum() * 10.0)             # GHz
    cu = ((hw >> 32) & 0xF) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xF)
    ids, per_cu = np.unique(cu, return_counts=True)
    busy = np.zeros(len(ids))
    for i, c in enumerate(ids):
        busy[i] = dur[cu == c].sum()
    start_off = (t0 - t0.min()) * 0.01
    print(f"{spec}: {len(s)} workgroups on {len(ids)} CUs; events {ms * 1e3:.1f} us, first start -> last end {span:.1f} us")
    print(f"   workgroup duration min/median/max {dur.min():.1f}/{np.median(dur):.1f}/{dur.max():.1f} us; "
          f"in-kernel clock {clk:.2f} GHz; workgroups per CU min/max {per_cu.min()}/{per_cu.max()}")
    print(f"   start offsets: median {np.median(start_off):.1f} us, 90 % {np.percentile(start_off, 90):.1f}, max {start_off.max():.1f}; "
          f"end offsets: 10 % {np.percentile((t1 - t0.min()) * 0.01, 10):.1f}, median {np.median((t1 - t0.min()) * 0.01):.1f} us")
    q = np.percentile(dur, [10, 25, 75, 90])
    print(f"   duration 10/25/75/90 %: {q[0]:.1f}/{q[1]:.1f}/{q[2]:.1f}/{q[3]:.1f} us; per-XCD mean duration: " +
          " ".join(f"{dur[((hw >> 32) & 0xF) == xc].mean():.1f}" for xc in range(8) if (((hw >> 32) & 0xF) == xc).any()))


def run_model(bs=16, n_streams=1):
    """Timeline of one forward's conv launches (the stamped kernels: conv_igemm incl. heads, conv3x3_halo; the stem,
    the fused unit, SPP are not stamped and show up as gaps).  Start / end of each launch = first workgroup start /
    last workgroup end on the 100 MHz clock all launches share."""
    import importlib.util
    import numpy as np
    import torch
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from pytorch_yolo_amd._lib import OP_CONV, OP_HEAD_DECODE
    from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict
    wl = bench.WORKLOADS["spp"]
    dev = torch.device("cuda", 0)
    model = wl["cls"](**wl["kw"]).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
    model = model.to(dev)
    model.n_streams = n_streams
    x = synth_images(bs * n_streams, 640, 640, 0).to(dev)
    plan = model.plan_for(x)
    io, ps = plan.new_outputs()
    for _ in range(3):
        plan._launch(x, io, ps)
    torch.cuda.synchronize()
    stride = 4 * 8192
    slots = 80 * n_streams
    stamps = torch.zeros(slots * stride, dtype=torch.int64, device=dev)
    os.environ["YOLO_STAMP_PTR"] = hex(stamps.data_ptr())
    os.environ["YOLO_STAMP_STRIDE"] = str(stride)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan._launch(x, io, ps)
    e1.record()
    torch.cuda.synchronize()
    del os.environ["YOLO_STAMP_PTR"]
    s = stamps.cpu().numpy().reshape(slots, 8192, 4)
    rows = []
    for i in range(slots):
        b = s[i][s[i][:, 1] > 0]
        if len(b):
            rows.append((i, len(b), b[:, 0].min(), b[:, 1].max(), np.median((b[:, 1] - b[:, 0]) * 0.01),
                         b[:, 2].sum() / ((b[:, 1] - b[:, 0]).sum() * 10.0)))
    t_first = min(r[2] for r in rows)
    print(f"bs {bs} x {n_streams} stream(s): {len(rows)} stamped launches, events {e0.elapsed_time(e1) * 1e3:.1f} us")
    print("  # blocks  start us   dur us  gap-before us  median block us  clock GHz")
    order = sorted(rows, key=lambda r: r[2])
    prev_end = None
    tot_dur = tot_gap = 0.0
    for i, nb, a, b, med, clk in order:
        gap = (a - prev_end) * 0.01 if prev_end is not None else 0.0
        print(f"{i:3d} {nb:6d} {(a - t_first) * 0.01:9.1f} {(b - a) * 0.01:8.1f} {gap:10.1f} {med:12.1f} {clk:10.2f}")
        tot_dur += (b - a) * 0.01
        if prev_end is not None and n_streams == 1:
            tot_gap += gap
        prev_end = b if prev_end is None else max(prev_end, b)
    print(f"sum of launch durations {tot_dur:.1f} us; sum of gaps (incl. the unstamped stem / fused unit / SPP) {tot_gap:.1f} us; "
          f"first start -> last end {(max(r[3] for r in rows) - t_first) * 0.01:.1f} us")


if __name__ == "__main__":
    if "--build" in sys.argv or "--build-all" in sys.argv:
        build()
        if "--build-all" not in sys.argv:
            sys.exit(0)
        build(["YOLO_ABL_NOBARRIER"], "nobarrier")
        build(["YOLO_ABL_NOLDS"], "nolds")
        build(["YOLO_ABL_NOBARRIER", "YOLO_ABL_NOLDS"], "nobarrier_nolds")
        sys.exit(0)
    args = sys.argv[1:]
    lib = LIB
    if args and args[0] == "--variant":       # nobarrier | nolds | nobarrier_nolds (timing only: results are wrong)
        lib = LIB.replace(".so", f"_{args[1]}.so")
        args = args[2:]
    if not os.path.exists(lib):
        raise SystemExit("run `python tools/block_timeline.py --build` first")
    os.environ["YOLO_HIP_LIB"] = lib
    sys.path.insert(0, ROOT)
    if args and args[0] == "--model":
        run_model(16, int(args[1]) if len(args) > 1 else 1)
        sys.exit(0)
    for spec in args or ["16,40,40,256,512,3,1,1", "16,80,80,128,256,3,1,1", "16,20,20,512,1024,3,1,1",
                         "16,80,80,256,128,1,1", "16,40,40,512,256,1,1"]:
        run(spec)
