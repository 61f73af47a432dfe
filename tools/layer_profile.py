#!/usr/bin/env python3
"""Per-layer timing of one forward (HIP events around each recorded op, median of R repeats).

    python tools/layer_profile.py --workload spp --bs 32 [--repeat 5]
Prints one row per launch: kind, GEMM view (M, N, K), ms, TFLOP/s, algorithmic GB/s.
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from pytorch_yolo_amd import kernels as K
from pytorch_yolo_amd._lib import (OP_CONV, OP_CONV1_NCHW, OP_DWCONV, OP_HEAD_DECODE, OP_MAXPOOL, OP_RESUNIT, OP_SPP,
                                   OP_STEM, YoloOp)
from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="spp")
    ap.add_argument("--bs", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--launch-cus", type=int, default=0, help="CU count the tile rules plan for (default: 256, or 128 under --cu-mask half)")
    ap.add_argument("--compact", action="store_true", help="heads in the compact NMS form (filter in the epilogue, no io store)")
    ap.add_argument("--no-p", action="store_true", help="heads skip the raw p store (what detect() / bench.py run)")
    ap.add_argument("--cu-mask", default="", help="'half': time the ops on a stream that owns half of every XCD's CUs (YOLO_CU_PARTITION=split)")
    args = ap.parse_args()
    wl = bench.WORKLOADS[args.workload]
    bs = args.bs or wl["bs"]
    dev = torch.device("cuda", 0)
    model = wl["cls"](**wl["kw"]).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
    model = model.to(dev)
    model.n_streams = 1
    x = synth_images(bs, wl["hw"], wl["hw"], 0).to(dev)
    if args.cu_mask == "half":
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        torch.cuda.synchronize()
        torch.cuda.set_stream(K.cu_masked_stream([b for b in range(n_cu) if (b // 8) < n_cu // 16], dev))
        K.set_launch_cus(args.launch_cus or n_cu // 2)
    elif args.launch_cus:
        K.set_launch_cus(args.launch_cus)
    plan = model.plan_for(x)
    plan.feed(x)
    if args.compact:
        cws = plan.compact_workspace()
        plan._bind_outputs(None, tuple(None for _ in plan.heads), (cws, bench.CONF_THRES, 2.0))
    else:
        plan._bind_outputs(*plan.new_outputs(want_p=not args.no_p))
    K.run_ops(plan.op_array, plan.n_ops)
    torch.cuda.synchronize()
    times = [[] for _ in range(plan.n_ops)]
    for _ in range(args.repeat):
        for i in range(plan.n_ops):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            one = C.cast(C.byref(plan.op_array, i * C.sizeof(YoloOp)), C.POINTER(YoloOp))
            e0.record()
            K.run_ops(one, 1)
            e1.record()
            e1.synchronize()
            times[i].append(e0.elapsed_time(e1))
    tot_ms = tot_fl = 0.0
    print(f"{'#':>3} {'kind':7} {'M':>9} {'N':>5} {'K':>5} {'k':>1} {'s':>1} {'ms':>8} {'TFLOP/s':>8} {'GB/s':>7}  flags")
    for i in range(plan.n_ops):
        op = plan.op_array[i]
        d = op.conv
        ms = statistics.median(times[i])
        tot_ms += ms
        if op.kind == OP_STEM:
            M = d.n * d.ho * d.wo
            fl = 2.0 * d.n * d.h * d.w * 32 * 27 + 2.0 * M * 64 * 288
            tot_fl += fl
            by = d.n * d.h * d.w * 3 * 4 + M * 64 * 2
            print(f"{i:3d} {'stem':7} {M:9d} {64:5d} {288:5d} 3 2 {ms:8.4f} {fl / ms / 1e9:8.1f} {by / ms / 1e6:7.0f}  conv1+s2")
        elif op.kind in (OP_CONV, OP_CONV1_NCHW, OP_HEAD_DECODE, 9, 11):
            M, N, Kd = d.n * d.ho * d.wo, d.cout, d.ksize * d.ksize * d.cin
            fl = 2.0 * M * N * Kd
            by = d.n * d.h * d.w * d.cin * 2 + M * N * (4 if d.out_dtype else 2) * (4 if d.upsample2x else 1) + N * Kd * 2
            if op.kind == OP_HEAD_DECODE:
                by += M * N * 4
            if op.residual:
                by += M * N * 2
            if op.y_aux:
                by += M * N * 2
            tot_fl += fl
            flags = ("head+decode " if op.kind == OP_HEAD_DECODE else "") + ("res " if op.residual else "") + ("aux " if op.y_aux else "") + ("up " if d.upsample2x else "") + ("f32" if d.out_dtype else "")
            print(f"{i:3d} {'conv':7} {M:9d} {N:5d} {Kd:5d} {d.ksize:1d} {d.stride:1d} {ms:8.4f} {fl / ms / 1e9:8.1f} {by / ms / 1e6:7.0f}  {flags}")
        elif op.kind == OP_RESUNIT:
            M, Cc = d.n * d.h * d.w, d.cout
            fl = 2.0 * M * (Cc * Cc // 2) * 10
            tot_fl += fl
            by = M * Cc * 2 * 3
            print(f"{i:3d} {'resunit':7} {M:9d} {Cc:5d} {Cc * 5:5d} 3 1 {ms:8.4f} {fl / ms / 1e9:8.1f} {by / ms / 1e6:7.0f}  {'aux' if op.y_aux else ''}")
        elif op.kind == 10:     # OP_MBCONV: hidden in the K column, bytes = x read + y written
            M, hid = d.n * d.ho * d.wo, op.kpad_pre
            fl = (2.0 * d.n * d.h * d.w * d.cin * hid if op.w_pre else 0.0) + 2.0 * M * hid * (9 + d.cout)
            tot_fl += fl
            by = d.n * d.h * d.w * d.cin * 2 + M * d.cout * 2
            print(f"{i:3d} {'mbconv':7} {M:9d} {d.cout:5d} {hid:5d} 3 {d.stride:1d} {ms:8.4f} {fl / ms / 1e9:8.1f} {by / ms / 1e6:7.0f}  {'res' if d.res_c_total else ''}")
        else:
            kind = {OP_MAXPOOL: "pool", OP_SPP: "spp", OP_DWCONV: "dwconv", 12: "shuffle", 15: "se"}.get(op.kind, f"op{op.kind}")
            by = d.n * d.h * d.w * d.cin * 2 * (4 if op.kind == OP_SPP else 3 if op.kind == 15 else 2)
            print(f"{i:3d} {kind:7} {d.n * d.h * d.w:9d} {d.cin:5d} {'':5} {d.ksize:1d} {d.stride:1d} {ms:8.4f} {'':8} {by / ms / 1e6:7.0f}")
    print(f"total {tot_ms:.3f} ms  conv {tot_fl / 1e12:.3f} TFLOP -> {tot_fl / tot_ms / 1e9:.1f} TFLOP/s over the per-op sum")


if __name__ == "__main__":
    main()
