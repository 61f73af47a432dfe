#!/usr/bin/env python3
"""Headline benchmark: images/s of detect() = forward + YOLO decode + MERGE-NMS on synthetic
640x640 batches, YOLOv3-SPP, bs=32 per MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...          # starts N ranks itself (one child process per GPU under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path over one batch of 32 images per GPU (inputs already
resident in HBM).  For N>1 every rank runs its own 32 images (weak scaling) and the per-rank
detections are all-gathered (RCCL) inside the timed step.  Rank 0 prints ONE JSON line.
`--gpus N` with fewer than N visible GPUs, or a WORLD_SIZE that disagrees with N, exits non-zero: a run never reports
fewer ranks than it was asked for.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pytorch_yolo_amd import (YOLOv3SPP, YOLOv3Tiny, YOLOv3TinyEfficient, YOLOv3TinyMobile, YOLOv3TinyShuffle,   # noqa: E402
                              YOLOv3TinySqueeze)
from pytorch_yolo_amd.distributed import PipelinedGather                    # noqa: E402
from pytorch_yolo_amd.utils.synthetic import calibrate_plain_heads, synth_images, synth_state_dict  # noqa: E402
from pytorch_yolo_amd.utils.utils import nms_capacity, nms_raw               # noqa: E402
from pytorch_yolo_amd import kernels as K                                    # noqa: E402

SPP_ANCHORS = (((10., 13.), (16., 30.), (33., 23.)), ((30., 61.), (62., 45.), (59., 119.)),
               ((116., 90.), (156., 198.), (373., 326.)))
# algorithmic conv work per image: 2 x MACs of the conv layers only
GFLOP_PER_IMG = {"spp": 156.73, "tiny": 5.565, "mobile": 2.572}
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0            # HBM3E spec (6.3 TB/s measured achievable, same guide)
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
    "efficient": dict(cls=YOLOv3TinyEfficient, kw=dict(), hw=416, bs=32,
                      name="YOLOv3-tiny EfficientNet-B0 416x416 bs=32/GPU detect()"),
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


def power_sample(step_fn, seconds: float = 2.0):
    """Socket power and shader clock while the timed workload keeps running (OUTSIDE the timed region): rocm-smi read about
    twice a second with ~0.3 s of steps queued ahead of every read.  DESIGN.md 3.2a: the MFMA-dense layers run against the
    socket power limit, which is what holds roofline.frac where it is.  None when rocm-smi is not there."""
    import re
    import shutil
    import statistics
    import subprocess
    if shutil.which("rocm-smi") is None:
        return None
    watts, mhz = [], []
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < seconds:
        t1 = time.perf_counter()
        for _ in range(60):
            step_fn(i)
            i += 1
        try:
            txt = subprocess.run(["rocm-smi", "-d", str(torch.cuda.current_device()), "--showpower", "--showclocks"],
                                 capture_output=True, text=True, timeout=10).stdout
        except Exception:                                        # noqa: BLE001 (diagnostic only)
            return None
        torch.cuda.synchronize()
        busy = time.perf_counter() - t1
        pw = re.search(r"Power \(W\): ([\d.]+)", txt)
        sc = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
        if pw and sc:
            watts.append(float(pw.group(1)))
            mhz.append(int(sc.group(1)))
    if not watts:
        return None
    return {"socket_w": round(statistics.median(watts), 1), "sclk_mhz": int(statistics.median(mhz)), "samples": len(watts),
            "cap_w": 1400, "how": "rocm-smi --showpower --showclocks while extra steps run after the timed region"}


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
                    "shuffle": (om.tiny_shuffle_forward, YOLOv3TinyShuffle),
                    "efficient": (om.tiny_efficient_forward, YOLOv3TinyEfficient)}[workload]
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


def _free_port() -> int:
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start N fresh rank processes (one per GPU) and relay rank 0's
    JSON line.  This parent never initialises the GPU (torch.cuda.device_count() does not) and never re-execs."""
    dry = args.dry_run
    if not dry:
        visible = torch.cuda.device_count()
        if visible < args.gpus:
            print(f"bench.py: --gpus {args.gpus} requested but only {visible} GPU(s) are visible; refusing to report a "
                  f"{args.gpus}-GPU number from fewer ranks", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world, rank):
    """Launcher / collective rehearsal without a GPU (tests/test_distributed_cpu.py): N ranks over gloo exchange detection
    buffers of the real shapes through the same gather and print the JSON line with value null.  Nothing is measured."""
    from pytorch_yolo_amd.distributed import gather_detections
    dist.init_process_group("gloo")
    bs, cap = 4, 64
    dets = torch.full((bs, cap, 7), float(rank), dtype=torch.float32)
    count = torch.full((bs,), rank + 1, dtype=torch.int32)
    for _ in range(args.warmup + args.steps):
        all_dets, all_count = gather_detections(dets, count, equal_shards=True)
    dist.barrier()
    ok = all_count.tolist() == [r + 1 for r in range(world) for _ in range(bs)] and all_dets.shape[0] == world * bs
    if rank == 0:
        print(json.dumps({"metric": "images/sec YOLOv3-SPP 640x640 bs=32, forward + decode + MERGE-NMS, pipelined batches", "value": None, "unit": "images/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                          "data": "dry-run: no GPU, launcher + gloo all-gather rehearsal only (nothing measured)",
                          "config": {"workload": "dry-run", "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                                     "gather_ok": bool(ok)}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def solo_kernel_table(plan0, repeat=5):
    """Every launch of ONE sub-batch list timed alone (HIP events on the launch stream, median of ``repeat``), grouped by
    what it computes; returns the groups sorted by total time.  'Solo' = nothing else on the chip: the concurrent-stream
    step time is NOT the sum of these."""
    import ctypes as C
    import statistics
    from pytorch_yolo_amd._lib import (OP_CONV, OP_CONV1_NCHW, OP_CONV1_POOL, OP_CONV_POOL, OP_HEAD_DECODE, OP_RESUNIT, OP_STEM,
                                       YoloOp)
    groups = {}
    for i in range(plan0.n_ops):
        op = plan0.op_array[i]
        d = op.conv
        one = C.cast(C.byref(plan0.op_array, i * C.sizeof(YoloOp)), C.POINTER(YoloOp))
        ts = []
        for _ in range(repeat):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            K.run_ops(one, 1)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = statistics.median(ts)
        # grouped by the KERNEL that runs (the name yolo_conv2d_pick reports for the shape), not by map size: the dominant kernel of
        # SPP-640 is one kernel on three map sizes (measurement nit 12)
        if op.kind == OP_STEM:
            label, fl = "stem2_kernel (conv 3x3 3->32 + 3x3/s2 32->64, one launch)", 2.0 * d.n * d.h * d.w * 32 * 27 + 2.0 * d.n * d.ho * d.wo * 64 * 288
        elif op.kind == OP_RESUNIT:
            label, fl = f"fused residual unit kernels (C = {d.cout})", 2.0 * d.n * d.h * d.w * (d.cout * d.cin) * 10
        elif op.kind == OP_HEAD_DECODE:
            label, fl = "conv_igemm_bf16_kernel<..., DECODE> (head conv + decode + row filter)", 2.0 * d.n * d.ho * d.wo * d.cout * d.ksize * d.ksize * d.cin
        elif op.kind == OP_CONV:
            fam = K.conv2d_pick(d, bool(op.residual), bool(op.y_aux)).split("<")[0]
            name = {"t20v2": "conv3x3_t20v2_kernel", "t20s2": "conv3x3s2_t20_kernel", "igemm": "conv_igemm_bf16_kernel", "stream1x1": "conv1x1_stream_kernel",
                    "halo": "conv3x3_halo_kernel"}.get(fam, fam)
            label, fl = f"{name} ({d.ksize}x{d.ksize}/s{d.stride})", 2.0 * d.n * d.ho * d.wo * d.cout * d.ksize * d.ksize * d.cin
        elif op.kind in (OP_CONV1_NCHW, OP_CONV1_POOL, OP_CONV_POOL):
            fl = 2.0 * d.n * d.ho * d.wo * d.cout * d.ksize * d.ksize * d.cin
            label = "first-layer / small-cin conv (+ pool) kernels (conv_small.hip)"
        else:
            label, fl = f"op kind {op.kind}", 0.0
        g = groups.setdefault(label, dict(label=label, launches=0, ms=0.0, flops=0.0))
        g["launches"] += 1
        g["ms"] += ms
        g["flops"] += fl
    return sorted(groups.values(), key=lambda g: -g["ms"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="spp", choices=list(WORKLOADS))
    ap.add_argument("--bs", type=int, default=0, help="images per GPU (default: the BASELINE config)")
    ap.add_argument("--hw", type=int, default=0, help="square input size (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="concurrent pipelines per GPU (1 = off)")
    ap.add_argument("--pipeline", default="batches", choices=["batches", "halves"],
                    help="what a pipeline carries: whole batches, successive steps alternating between the pipelines (default), or "
                         "one sub-batch of every step each")
    ap.add_argument("--cu-partition", action="store_true",
                    help="give each pipeline half of every XCD's CUs (hipExtStreamCreateWithCUMask) and size its grids for them: the "
                         "default of rounds 2-3 (+1.0 %% then); since the compact NMS form and the high-priority NMS stream of round 4 "
                         "the shared chip is as fast or faster (6,604 vs 6,593 and 6,545 vs 6,459 images/s on two boxes), so it is off")
    ap.add_argument("--materialize-io", action="store_true",
                    help="store io and run the plain NMS on it (A/B of the compact NMS form, which is the default: the heads filter "
                         "their own rows and detect() never writes io)")
    ap.add_argument("--no-api", action="store_true", help="skip the API-level model.detect() timing")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2.5 s sustained-rate window after the timed region")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: rehearse the N-rank launcher and the all-gather over gloo, print a line with value null")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "RANK" not in os.environ and (args.gpus > 1 or args.dry_run):
        sys.exit(launch_ranks(args, sys.argv[1:]))               # BEFORE any GPU call in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` or under "
                         f"torch.distributed.run with --nproc-per-node {args.gpus}")
    if args.dry_run:
        sys.exit(dry_run(args, world, rank))
    # rehearsal on a one-GPU box only: YOLO_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo (RCCL refuses two
    # ranks on one device); the numbers of such a run mean nothing, it exercises the sharded control flow
    rehearse = os.environ.get("YOLO_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the sharded path runs even with one rank, so that it can be rehearsed
    # on a single GPU; the plain `python bench.py` of the N=1 contract stays collective-free
    sharded = world > 1 or ("RANK" in os.environ and os.environ.get("YOLO_BENCH_SHARDED_AT_1") == "1")
    backend = None
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        backend = dist.get_backend()
        assert dist.get_world_size() == args.gpus

    wl = dict(WORKLOADS[args.workload])
    if args.hw and args.hw != wl["hw"]:
        wl["name"] = wl["name"].replace(f"{wl['hw']}x{wl['hw']}", f"{args.hw}x{args.hw}") + " (NOT the BASELINE input size)"
        wl["hw"] = args.hw
    bs = args.bs or wl["bs"]
    model = wl["cls"](**wl["kw"]).eval()
    model.load_state_dict(synth_state_dict(model.state_dict(), 1234, n_class=80))
    model = model.to(dev)
    model.n_streams = args.streams
    x = synth_images(bs, wl["hw"], wl["hw"], seed=rank).to(dev)      # per-rank images, resident in HBM
    head_gain = None
    if args.workload != "spp":          # plain-conv heads on another encoder: give the NMS leg a realistic load (synthetic weights only)
        head_gain = calibrate_plain_heads(model, x)

    plan = model.plan_for(x)
    rows, nc = plan.rows_total, model.n_class
    cap = nms_capacity(rows, nc)
    def new_nms_out():
        return (torch.empty((bs, cap, 7), dtype=torch.float32, device=dev),
                torch.empty((bs, cap), dtype=torch.int32, device=dev),
                torch.empty((bs,), dtype=torch.int32, device=dev))
    nms_out = new_nms_out()
    # detect() = non_max_suppression(model(x)[0], ...) (reference utils/utils.py:374-378): the raw head tensors p, which forward()
    # also returns, are not part of it - the head kernels decode in their epilogue and skip the p store (as model.detect() does)
    io, ps = plan.new_outputs(want_p=False)
    flops_step = plan.conv_flops()

    n_streams = plan.n_streams
    # whole-batch pipelines: step i goes down pipeline i % S as ONE launch list of all its images, S batches in flight, each with
    # its own output buffers (engine.StreamedPlan.launch_detect(whole_batch=True))
    whole = args.pipeline == "batches" and n_streams > 1
    sets = [(io, ps, nms_out)] + [(plan.new_outputs(want_p=False) + (new_nms_out(),)) for _ in range(n_streams - 1 if whole else 0)]
    total_steps = args.steps + args.warmup
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_streams)]
          for _ in range(total_steps)]
    # no per-step join: the S sub-batch pipelines run freely until the final sync; when sharded, the detections
    # all-gather of every step rides a side stream (distributed.PipelinedGather) instead of joining them
    gatherer = PipelinedGather(bs, cap, n_streams, dev) if sharded else None

    calls = [0]

    def step(i, timed=True):
        """Per stream: conv1 (reads the NCHW batch) -> 75 conv launches + SPP -> 3 decodes -> NMS on its sub-batch
        (-> join + all-gather when sharded over ranks).  HIP events bracket every stream's conv launch list on
        the stream it is launched on."""
        io, ps, nms_out = sets[calls[0] % len(sets)]         # call k goes down pipeline k % S: its buffer set
        calls[0] += 1
        tm = ev[i] if timed else None
        if gatherer is None:
            plan.launch_detect(x, io, ps, nms_out, CONF_THRES, NMS_THRES, timing=tm, join=False, whole_batch=whole,
                               cu_partition=args.cu_partition, compact=not args.materialize_io)
            return nms_out[0], nms_out[2]
        plan.launch_detect(x, io, ps, nms_out, CONF_THRES, NMS_THRES, timing=tm, join=False, whole_batch=whole,
                           after_nms=gatherer.begin(nms_out), compact=not args.materialize_io)
        all_dets, all_count, _ = gatherer.exchange()
        return all_dets, all_count

    def sync_all():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    # (measured BEFORE the free-running loop: that loop creates the CU-partitioned streams, blocking streams whose mere existence
    # taxes every NULL-stream operation of a host-bound detect() loop - engine.StreamedPlan)
    # API-level number beside the device-level one: model.detect() as a caller sees it (fresh output tensors every call, the
    # count D2H copy and the list[Tensor | None] split inside the timed region, host-synchronous); sharded runs time
    # detect_sharded (forward + NMS + all-gather + split on every rank)
    api_ips = stream_ips = None
    if not args.no_api:
        from pytorch_yolo_amd.distributed import detect_sharded
        k_api = max(3, min(args.steps, 20))
        with torch.no_grad():
            call = (lambda: detect_sharded(model, x, CONF_THRES, NMS_THRES, equal_shards=True)) if sharded else (lambda: model.detect(x, CONF_THRES, NMS_THRES))
            call()
            sync_all()
            ta = time.perf_counter()
            for _ in range(k_api):
                res = call()
            sync_all()
            tb = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=dev)
        if sharded:
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        api_ips = round(bs * world * k_api / float(tb.item()), 2)
        assert len(res) == bs * world
        if not sharded and os.environ.get("YOLO_BENCH_SKIP_STREAM_API") != "1":          # the same API pipelined: model.detect_stream() (two batches in flight, lists handed out one batch late)
            with torch.no_grad():
                for _ in model.detect_stream((x for _ in range(4)), CONF_THRES, NMS_THRES):
                    pass
                sync_all()
                # at least 0.25 s of batches (the small models run 0.4 ms per batch: 40 batches would time the generator's start-up)
                n_stream = max(2 * k_api, int(0.25 * api_ips / bs) + 1)
                ta = time.perf_counter()
                n_out = sum(len(r) for r in model.detect_stream((x for _ in range(n_stream)), CONF_THRES, NMS_THRES))
                sync_all()
                stream_ips = round(n_out / (time.perf_counter() - ta), 2)
            assert n_out == bs * n_stream

    with torch.no_grad():
        for i in range(args.warmup):
            step(i)
        sync_all()
        t0 = time.perf_counter()
        for i in range(args.warmup, total_steps):
            dets, counts = step(i)
        sync_all()
        dt = time.perf_counter() - t0
    # conv-family time of a step, measured: HIP events bracket every launch list on the stream it is launched on; the lists of
    # successive steps (whole-batch pipelines) or of the sub-batches of one step run side by side, so the figure is the SPAN from the
    # first timed list's start to the last timed lists' end divided by the timed steps - no assumption about how well the
    # pipelines overlap (the former "one list's time / pipelines" overstated the rate whenever they serialise)
    first = ev[args.warmup][0][0]
    tail = range(max(args.warmup, total_steps - n_streams), total_steps)
    span_ms = max(first.elapsed_time(e1) for i in tail for (_, e1) in (ev[i][:1] if whole else ev[i]))
    conv_ms = [span_ms / args.steps]
    list_ms = [ev[i][0][0].elapsed_time(ev[i][0][1]) for i in range(args.warmup, total_steps)]      # one list, start to end
    n_dets = counts.cpu().tolist()

    # the same steps over a window of >= 2.5 s (the timed region of `--steps 20` is 0.1 s on a chip whose clock follows a power cap)
    sustained_ips = None
    if world == 1 and not args.no_sustained:
        n_sus = max(args.steps, int(2.5 / (dt / args.steps)) + 1)
        with torch.no_grad():
            sync_all()
            ts = time.perf_counter()
            for i in range(n_sus):
                step(args.warmup + i % args.steps, timed=False)
            sync_all()
        sustained_ips = round(bs * n_sus / (time.perf_counter() - ts), 2)

    # the reference-precision mode (model.precision = "fp32": float32 activations and weights on the f32 MFMA, the mode whose NMS
    # kept-index sets equal the reference's end to end): detect() through the API, a few calls
    fp32_ips = None
    if world == 1 and not args.no_api and args.workload in ("spp", "tiny"):
        m32 = wl["cls"](**wl["kw"]).eval()
        m32.load_state_dict(synth_state_dict(m32.state_dict(), 1234, n_class=80))
        m32 = m32.to(dev)
        m32.precision = "fp32"
        if head_gain is not None:
            calibrate_plain_heads(m32, x)
        with torch.no_grad():
            m32.detect(x, CONF_THRES, NMS_THRES)
            sync_all()
            ts = time.perf_counter()
            for _ in range(3):
                m32.detect(x, CONF_THRES, NMS_THRES)
            sync_all()
        fp32_ips = round(3 * bs / (time.perf_counter() - ts), 2)
        del m32

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if sharded:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    traffic, traffic_src = None, None
    try:        # HBM bytes of the conv launch list per step: from committed rocprofv3 --pmc passes of this command, not live
        tj = json.load(open(os.path.join(ROOT, "profiles", "conv_traffic.json")))[args.workload]
        if tj["images_per_gpu"] == bs:
            traffic, traffic_src = tj["hbm_bytes_per_step"], tj.get("source", "profiles/conv_traffic.json") + " (rocprofv3 --pmc passes, NOT measured in this run)"
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        total_imgs = bs * world * args.steps
        conv_ms_avg = sum(conv_ms) / len(conv_ms)
        achieved = flops_step / (conv_ms_avg * 1e-3) / 1e12
        cfg = {"workload": wl["name"], "images_per_gpu": bs, "global_batch": bs * world,
               "n_class": nc, "conf_thres": CONF_THRES, "nms_thres": NMS_THRES,
               "sharding": f"batch x{world}" + (" + RCCL all-gather of detections (side stream)" if sharded else ""),
               "mean_detections_per_image": round(sum(n_dets) / max(1, len(n_dets)), 1),
               "streams_per_gpu": n_streams, "precision": model.precision}
        cfg["raw_head_tensors_p"] = "not materialised: detect() discards them (forward() stores them; tests cover both)"
        cfg["decoded_rows_io"] = ("materialised (--materialize-io): the plain NMS reads them back" if args.materialize_io else
                                  "not materialised: the head epilogues filter their own rows into the NMS workspace (compact form); forward() stores them")
        if n_streams > 1:
            cfg["pipelines"] = (f"{n_streams} x whole batches of {bs} (successive steps alternate)" if whole
                                else f"{n_streams} x sub-batches of {bs // n_streams} of every step")
        cfg["hip_hw_queues"] = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        if n_streams > 1:
            used = (plan._full_streams or plan.streams) if whole else plan.pipe_streams
            cfg["cu_partition"] = "half of every XCD per stream" if type(used[0]).__name__ == "ExternalStream" else "off (streams share the chip)"
        if sharded:
            cfg["rccl_ranks"], cfg["backend"] = dist.get_world_size(), backend
            try:
                cfg["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:                                    # noqa: BLE001 (version query only)
                pass
        if api_ips is not None:
            cfg["detect_api_images_per_s"] = api_ips
        if stream_ips is not None:
            cfg["detect_stream_api_images_per_s"] = stream_ips
        if sustained_ips is not None:
            cfg["sustained_images_per_s"] = sustained_ips          # >= 2.5 s of the same steps after the timed region
        if fp32_ips is not None:
            cfg["fp32_mode_images_per_s"] = fp32_ips               # model.precision = "fp32" through detect(): the exact-kept-set mode
        if head_gain is not None:
            cfg["synthetic_head_gain"] = round(head_gain, 3)
        out = {
            # what `value` times: the queued launch loop of the path (forward + decode + MERGE-NMS per 32 resident images, successive
            # batches alternating between two pipelines, no host sync inside the timed region) - the rate a serving loop can reach;
            # the same path through the drop-in API, host-synchronous per batch, is config.detect_api_images_per_s, and through the
            # pipelined API generator config.detect_stream_api_images_per_s
            "metric": ("images/sec YOLOv3-SPP 640x640 bs=32, forward + decode + MERGE-NMS, pipelined batches" if args.workload == "spp"
                       else f"images/sec {wl['name']}, pipelined batches"),
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
            "config": cfg,
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": (plan if not hasattr(plan, "_full") or plan._full is None else plan._full[0]).algorithmic_bytes(detect=not args.materialize_io),
                         "kernel": "conv-family launch lists of the forward (stem, fused residual units, conv3x3_t20v2, conv3x3s2_t20, conv_igemm_bf16 / "
                                   "conv1x1_stream incl. head+decode); per step: HIP-event span from the first timed list's start to the last "
                                   "lists' end / timed steps (the pipelines' lists overlap; nothing is assumed about how well)",
                         "flops_per_step": flops_step, "ms_per_step_conv": round(conv_ms_avg, 4),
                         "floor_ms": round(max(flops_step / 1.25e15, 0.0) * 1e3, 4),      # 5.015 TFLOP at the 1.25 PFLOP/s a bare bf16 MFMA loop holds at the power cap
                         "ms_one_list_start_to_end": round(sum(list_ms) / len(list_ms), 4)},
        }
        if args.workload != "spp":
            # The small models are not MFMA work: 13..80 launches of 10..100 us whose bytes, not FLOPs, bound them.  Their line is
            # priced against HBM: algorithmic bytes of the launch list (engine.Plan.algorithmic_bytes) over its event time.
            abytes = plan.algorithmic_bytes(detect=not args.materialize_io)
            gbs = abytes / (conv_ms_avg * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                               "traffic": traffic, "traffic_source": traffic_src,
                               "kernel": "layer launch lists of the forward (conv / pool / depthwise / fused blocks incl. head+decode); per step: HIP-event span over the timed lists / timed steps",
                               "algorithmic_bytes_per_step": abytes, "ms_per_step_layers": round(conv_ms_avg, 4),
                               # what the list would take at the practical roofs, whichever binds (6.3 TB/s achievable HBM, 1.25 PFLOP/s at the
                               # power cap: MI355X_MICROARCH.md) - unlike `frac` this does not FALL when a fusion removes bytes from the numerator
                               "floor_ms": round(max(abytes / 6.3e12, flops_step / 1.25e15) * 1e3, 4),
                               "time_over_floor": round(conv_ms_avg / (max(abytes / 6.3e12, flops_step / 1.25e15) * 1e3), 2),
                               "mfma_tflops": round(achieved, 2), "mfma_frac": round(achieved / PEAK_BF16_TFLOPS, 4)}
        if world == 1:
            # the dominant kernel family by itself, live: every launch of one sub-batch list timed alone with HIP events
            with torch.no_grad():
                tab = solo_kernel_table((plan._full[0] if whole else plan.subs[0]) if hasattr(plan, "subs") else plan)
            top = tab[0]
            tf = top["flops"] / (top["ms"] * 1e-3) / 1e12
            out["roofline"]["dominant_kernel"] = {
                "label": top["label"], "launches_per_list": top["launches"], "avg_ms_solo": round(top["ms"] / top["launches"], 4),
                "flops_per_launch": top["flops"] / top["launches"], "achieved": round(tf, 1), "frac": round(tf / PEAK_BF16_TFLOPS, 4),
                "share_of_list_time": round(top["ms"] / sum(g["ms"] for g in tab), 3)}
        if world == 1 and not args.no_api:
            with torch.no_grad():
                pw = power_sample(lambda i: step(args.warmup + i % args.steps, timed=False))
            sync_all()
            if pw is not None:
                out["roofline"]["power"] = pw
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
