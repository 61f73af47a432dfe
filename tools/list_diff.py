#!/usr/bin/env python3
"""Launch-by-launch difference between two launch lists of the same model on the same images: the 16-image / 256-CU list that
model(x) runs and the 32-image / 128-CU list that bench.py times (test_benched_launch_list_bs32_whole_batch, item d).

    python tools/list_diff.py            # SPP-640: prints, per launch, the kernels the two lists pick and the relative rms / max
                                         # difference of the launch's output over the first 16 images
A launch whose inputs are bit-equal and whose picks add the K products in the same order must give bit-equal outputs."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _cases as CS
from helpers import build_case
from pytorch_yolo_amd import engine, kernels as K
from pytorch_yolo_amd._lib import OP_CONV, OP_HEAD_DECODE, OP_RESUNIT, YoloOp
from pytorch_yolo_amd.utils.synthetic import synth_images


def out_sym(nd):
    if nd.kind == "conv":
        return nd.attrs.get("up_into") or nd.attrs.get("pool_into") or nd.outs[0]
    return nd.outs[0]


def main():
    dev = torch.device("cuda", 0)
    model, sd, _ = build_case(CS.FULL_CASES["spp_640"])
    model = model.to(dev)
    x = torch.cat([synth_images(1, 640, 640, i) for i in range(32)], 0).to(dev)

    def make(bs):
        rec = engine.Recorder(bs, 3, 640, 640)
        model._trace(rec, rec.input)
        return engine.Plan(rec, dev, 80, 640)
    pa, pb = make(16), make(32)
    xa = x[:16].contiguous()
    pa.feed(xa), pb.feed(x)
    ioa, psa = pa.new_outputs()
    iob, psb = pb.new_outputs()
    pa._bind_outputs(ioa, psa), pb._bind_outputs(iob, psb)
    assert pa.n_ops == pb.n_ops
    for i in range(pa.n_ops):
        names = []
        for plan, cus in ((pa, 256), (pb, 128)):
            old = K.set_launch_cus(cus)
            op = plan.op_array[i]
            names.append(K.conv2d_pick(op.conv, bool(op.residual), bool(op.y_aux)).split(" grid")[0] if op.kind == OP_CONV else f"kind {op.kind}")
            K.run_ops(C.cast(C.byref(plan.op_array, i * C.sizeof(YoloOp)), C.POINTER(YoloOp)), 1)
            K.set_launch_cus(old)
        torch.cuda.synchronize()
        if pa.op_array[i].kind == OP_HEAD_DECODE:
            k = [h["op"] for h in pa.heads].index(i)
            a, b = psa[k].double(), psb[k][:16].double()
        else:
            sa, sb = out_sym(pa.op_nodes[i]), out_sym(pb.op_nodes[i])
            a = sa.buf.tensor[..., sa.c_offset:sa.c_offset + sa.c].double()
            b = sb.buf.tensor[:16, ..., sb.c_offset:sb.c_offset + sb.c].double()
        d = (a - b)
        rel = float(d.norm() / a.norm().clamp_min(1e-30))
        nz = float((d != 0).double().mean())
        print(f"{i:3d} {names[0]:58s} | {names[1] if names[1] != names[0] else '=':58s} rel rms {rel:.2e}  differing {nz:.4f}  max {float(d.abs().max()):.3g}")


if __name__ == "__main__":
    main()
