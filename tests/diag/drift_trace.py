#!/usr/bin/env python3
"""Per-layer drift trace of the HIP bf16 path on the GPU (run with gpurun): every launch of the recorded list is run one at a
time and its output tensor is compared with
  (a) the fp32 oracle (= the reference) and
  (b) the oracle under the bf16 rounding policy (oracle/policy.py = the fast path's own specification),
as relative rms error; column (c) is what the CPU model predicts for (a) (policy vs fp32).  If (b) stays at the
accumulation-order / double-rounding level while (a) tracks (c), the HIP path does exactly what its specification says and
the whole distance to the reference is the bf16 operand rounding.

    python tests/diag/drift_trace.py [--workload spp|tiny] [--out profiles/r02_drift_trace_spp640.md]
"""
import argparse
import ctypes as C
import os
import re
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel_rms(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="spp", choices=["spp", "tiny"])
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import _cases as Cs
    from helpers import build_case
    from oracle import models as om
    from oracle.policy import run_policy
    from pytorch_yolo_amd import kernels as K
    from pytorch_yolo_amd._lib import YoloOp
    from pytorch_yolo_amd.engine import _sym_to_nchw

    case = Cs.FULL_CASES["spp_640" if args.workload == "spp" else "tiny_416"]
    model, sd, x = build_case(case)
    fwd = om.spp_forward if args.workload == "spp" else om.tiny_forward
    anchors, nc = case[1]["anchors"], case[1]["n_class"]
    t32, t16 = {}, {}
    torch.set_num_threads(16)
    io32, p32 = run_policy(fwd, sd, x, anchors, nc, policy="fp32", taps=t32)
    io16, p16 = run_policy(fwd, sd, x, anchors, nc, policy="bf16", taps=t16)

    dev = torch.device("cuda", 0)
    model = model.to(dev)
    model.n_streams = 1
    xd = x.to(dev)
    plan = model.plan_for(xd)
    io, ps = plan.new_outputs()
    plan.feed(xd)
    plan._bind_outputs(io, ps)
    lines = ["| # | block | launch | rel rms vs fp32 reference | rel rms vs bf16-policy oracle | CPU model: policy vs fp32 |", "|---|---|---|---|---|---|"]
    head_i = 0
    for i in range(plan.n_ops):
        one = C.cast(C.byref(plan.op_array, i * C.sizeof(YoloOp)), C.POINTER(YoloOp))
        K.run_ops(one, 1)
        torch.cuda.synchronize()
        nd = plan.op_nodes[i]
        name = nd.attrs.get("name") if nd.kind == "conv" else None
        if name is None:
            lines.append(f"| {i} | ({nd.kind}) | - | | | |")
            continue
        kind = ("stem" if "stem_pre" in nd.attrs else "resunit" if "fuse_pre" in nd.attrs else "head+decode" if nd.attrs.get("head_fused")
                else "conv")
        key = name
        if nd.attrs["has_res"]:                                    # the launch stores x + conv: compare with the Add's output
            m = re.match(r"(down\d+)\.seq(\d+)\.1$", name)
            key = f"{m.group(1)}.add{m.group(2)}"
        r32, r16 = t32[key], t16[key]
        if nd.attrs.get("head_fused"):
            hd = plan.heads[head_i]
            got = ps[plan.heads.index(next(h for h in plan.heads if h["op"] == i))].cpu()       # [bs,na,ny,nx,no] raw logits
            bs, na, ny, nx, no = got.shape
            conv = lambda t: t.view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2)
            r32, r16 = conv(r32), conv(r16)
            head_i += 1
        else:
            dst = nd.attrs.get("pool_into") or nd.attrs.get("up_into") or nd.outs[0]
            got = _sym_to_nchw(dst).cpu()
            if nd.attrs.get("up_into") is not None:
                r32, r16 = (F.interpolate(t, scale_factor=2, mode="nearest") for t in (r32, r16))
            elif nd.attrs.get("pool_into") is not None:
                r32, r16 = (F.max_pool2d(t, 2, 2) for t in (r32, r16))
        lines.append(f"| {i} | {key} | {kind} | {rel_rms(got, r32):.5f} | {rel_rms(got, r16):.5f} | {rel_rms(r16, r32):.5f} |")
    torch.cuda.synchronize()
    plan._decode_unfused(io, ps)
    torch.cuda.synchronize()
    ioc = io.cpu()
    lines.append("")
    for tag, ref in (("fp32 reference", io32), ("bf16-policy oracle", io16)):
        box = (ioc[..., :4] - ref[..., :4]).abs()
        sc = (ioc[..., 4:] - ref[..., 4:]).abs()
        lines.append(f"io vs {tag}: max box err {float(box.max()):.4f} px, max score err {float(sc.max()):.5f}, rms score err "
                     f"{float(sc.double().pow(2).mean().sqrt()):.6f}")
    box = (io16[..., :4] - io32[..., :4]).abs()
    sc = (io16[..., 4:] - io32[..., 4:]).abs()
    lines.append(f"CPU model (bf16-policy oracle vs fp32 reference): max box err {float(box.max()):.4f} px, max score err {float(sc.max()):.5f}, "
                 f"rms score err {float(sc.double().pow(2).mean().sqrt()):.6f}")
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)) or ".", exist_ok=True)
        open(os.path.join(ROOT, args.out), "w").write(txt + "\n")


if __name__ == "__main__":
    main()
