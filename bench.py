#!/usr/bin/env python3
"""Headline benchmark: images/s of detect() = forward + YOLO decode + MERGE-NMS on synthetic
640x640 batches, YOLOv3-SPP, bs=32 per MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path over one batch of 32 images per GPU (inputs already
resident in HBM).  For N>1 every rank runs its own 32 images (weak scaling) and the per-rank
detections are all-gathered (RCCL) inside the timed step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pytorch_yolo_amd import YOLOv3SPP, YOLOv3Tiny, YOLOv3TinyMobile, YOLOv3TinyShuffle, YOLOv3TinySqueeze          # noqa: E402
from pytorch_yolo_amd.distributed import PipelinedGather                    # noqa: E402
from pytorch_yolo_amd.utils.synthetic import synth_images, synth_state_dict  # noqa: E402
from pytorch_yolo_amd.utils.utils import nms_capacity, nms_raw               # noqa: E402
from pytorch_yolo_amd import kernels as K                                    # noqa: E402

SPP_ANCHORS = (((10., 13.), (16., 30.), (33., 23.)), ((30., 61.), (62., 45.), (59., 119.)),
               ((116., 90.), (156., 198.), (373., 326.)))
# algorithmic conv work per image: 2 x MACs of the conv layers only
GFLOP_PER_IMG = {"spp": 156.73, "tiny": 5.565, "mobile": 2.572}
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
WORKLOADS = {
    "spp": dict(cls=YOLOv3SPP, kw=dict(anchors=SPP_ANCHORS), hw=640, bs=32,
                name="YOLOv3-SPP Darknet74 640x640 bs=32/GPU detect() = forward+decode+MERGE-NMS"),
    "tiny": dict(cls=YOLOv3Tiny, kw=dict(), hw=416, bs=32,
                 name="YOLOv3-tiny Darknet-15 416x416 bs=32/GPU detect()"),
    "mobile": dict(cls=YOLOv3TinyMobile, kw=dict(), hw=416, bs=64,
                   name="YOLOv3-tiny MobileNetV2 416x416 bs=64/GPU detect()"),
    "squeeze": dict(cls=YOLOv3TinySqueeze, kw=dict(), hw=416, bs=32,
                    name="YOLOv3-tiny SqueezeNet 1.1 416x416 bs=32/GPU detect()"),
    "shuffle": dict(cls=YOLOv3TinyShuffle, kw=dict(), hw=416, bs=32,
                    name="YOLOv3-tiny ShuffleNetV2 x1.0 416x416 bs=32/GPU detect()"),
}
CONF_THRES, NMS_THRES = 0.1, 0.5


def host_cores() -> int:
    """Threads the CPU baseline may use: the cgroup CPU quota if there is one, else the affinity
    mask, capped at 16 (a 1-GPU box's CPU share; more threads than that only thrash)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(workload: str, seconds_budget: float = 20.0):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores."""
    import numpy as np
    from oracle import models as om
    from oracle import nms as onms
    if workload == "spp":
        fwd, anchors, hw = om.spp_forward, om.SPP_ANCHORS, 640
        tmpl = YOLOv3SPP(anchors=SPP_ANCHORS).state_dict()
    else:
        fwd, cls = {"tiny": (om.tiny_forward, YOLOv3Tiny), "mobile": (om.tiny_mobile_forward, YOLOv3TinyMobile),
                    "squeeze": (om.tiny_squeeze_forward, YOLOv3TinySqueeze),
                    "shuffle": (om.tiny_shuffle_forward, YOLOv3TinyShuffle)}[workload]
        anchors, hw = om.TINY_ANCHORS, 416
        tmpl = cls().state_dict()
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = synth_state_dict(tmpl, 1234, n_class=80)
    bs = 4
    x = synth_images(bs, hw, hw, 0)
    with torch.no_grad():
        fwd(sd, x[:1], anchors, 80)                      # warm-up (thread pool, MKLDNN primitives)
        t0 = time.time()
        iters = 0
        while True:
            io, _ = fwd(sd, x, anchors, 80)
            onms.non_max_suppression(io.numpy(), CONF_THRES, NMS_THRES)
            iters += 1
            if time.time() - t0 > seconds_budget or (iters >= 3 and time.time() - t0 > 10.0):
                break
    dt = time.time() - t0
    return dict(value=round(bs * iters / dt, 3), unit="images/s", cores=cores, kind="port",
                sample=f"oracle fp32 torch forward + numpy MERGE-NMS, {iters} x bs={bs} {hw}x{hw}, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="spp", choices=list(WORKLOADS))
    ap.add_argument("--bs", type=int, default=0, help="images per GPU (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="concurrent sub-batch streams per GPU (1 = off)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal on a one-GPU box only: YOLO_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two
    # ranks on one device); the numbers of such a run mean nothing, it exercises the sharded control flow
    rehearse = os.environ.get("YOLO_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the sharded path runs even with one rank, so that it can be rehearsed
    # on a single GPU; the plain `python bench.py` of the N=1 contract stays collective-free
    sharded = world > 1 or ("RANK" in os.environ and os.environ.get("YOLO_BENCH_SHARDED_AT_1") == "1")
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    wl = WORKLOADS[args.workload]
    bs = args.bs or wl["bs"]
    model = wl["cls"](**wl["kw"]).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
    model = model.to(dev)
    model.n_streams = args.streams
    x = synth_images(bs, wl["hw"], wl["hw"], seed=rank).to(dev)      # per-rank images, resident in HBM

    plan = model.plan_for(x)
    rows, nc = plan.rows_total, model.n_class
    cap = nms_capacity(rows, nc)
    nms_out = (torch.empty((bs, cap, 7), dtype=torch.float32, device=dev),
               torch.empty((bs, cap), dtype=torch.int32, device=dev),
               torch.empty((bs,), dtype=torch.int32, device=dev))
    io, ps = plan.new_outputs()
    flops_step = plan.conv_flops()

    n_streams = plan.n_streams
    total_steps = args.steps + args.warmup
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_streams)]
          for _ in range(total_steps)]
    # no per-step join: the S sub-batch pipelines run freely until the final sync; when sharded, the detections
    # all-gather of every step rides a side stream (distributed.PipelinedGather) instead of joining them
    gatherer = PipelinedGather(bs, cap, n_streams, dev) if sharded else None

    def step(i):
        """Per stream: conv1 (reads the NCHW batch) -> 75 conv launches + SPP -> 3 decodes -> NMS on its sub-batch
        (-> join + all-gather when sharded over ranks).  HIP events bracket every stream's conv launch list on
        the stream it is launched on."""
        if gatherer is None:
            plan.launch_detect(x, io, ps, nms_out, CONF_THRES, NMS_THRES, timing=ev[i], join=False)
            return nms_out[0], nms_out[2]
        plan.launch_detect(x, io, ps, nms_out, CONF_THRES, NMS_THRES, timing=ev[i], join=False,
                           after_nms=gatherer.begin(nms_out))
        all_dets, all_count, _ = gatherer.exchange()
        return all_dets, all_count

    def sync_all():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.no_grad():
        for i in range(args.warmup):
            step(i)
        sync_all()
        t0 = time.perf_counter()
        for i in range(args.warmup, total_steps):
            dets, counts = step(i)
        sync_all()
        dt = time.perf_counter() - t0
    # conv time of a step: every stream's launch list (its share of the batch) is bracketed by HIP events on that
    # stream; the lists run concurrently, so the step's conv-family time is the LONGEST of them (the free-running
    # pipelines drift against each other, so a union over streams would mix work of neighbouring steps)
    conv_ms = []
    for i in range(args.warmup, total_steps):
        conv_ms.append(max(e0.elapsed_time(e1) for e0, e1 in ev[i]))
    n_dets = counts.cpu().tolist()

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if sharded:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    traffic = None
    try:        # HBM bytes of the conv launch list per step, from the committed rocprofv3 --pmc passes (cannot be read live)
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_conv_traffic.json")))
        if tj["workload"] == args.workload and tj["images_per_gpu"] == bs:
            traffic = tj["hbm_bytes_per_step"]
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        total_imgs = bs * world * args.steps
        conv_ms_avg = sum(conv_ms) / len(conv_ms)
        achieved = flops_step / (conv_ms_avg * 1e-3) / 1e12
        out = {
            "metric": "images/sec YOLOv3-SPP 640x640 bs=32 detect()" if args.workload == "spp" else f"images/sec {wl['name']}",
            "value": round(total_imgs / dt_max, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": wl["name"], "images_per_gpu": bs, "global_batch": bs * world,
                       "n_class": nc, "conf_thres": CONF_THRES, "nms_thres": NMS_THRES,
                       "sharding": f"batch x{world}" + (" + RCCL all-gather of detections (side stream)" if sharded else ""),
                       "mean_detections_per_image": round(sum(n_dets) / max(1, len(n_dets)), 1),
                       "streams_per_gpu": n_streams},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "kernel": "conv-family launch list of one forward (stem, resunit, conv_igemm_bf16 incl. head+decode, conv3x3_halo); per step the longest of the concurrent per-stream lists",
                         "flops_per_step": flops_step, "ms_per_step_conv": round(conv_ms_avg, 4)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
