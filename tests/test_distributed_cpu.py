"""world_size-2 gloo test of the only collective on the path: the all-gather of per-rank detections."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_yolo_amd.distributed import gather_detections, shard_bounds, split_gathered
    bs_local, cap = 3, 16
    g = torch.Generator().manual_seed(100 + rank)
    dets = torch.rand(bs_local, cap, 7, generator=g)
    count = torch.tensor([2 + rank, 0, 5], dtype=torch.int32)
    all_dets, all_count = gather_detections(dets, count, gather_cap=8)
    out = split_gathered(all_dets, all_count)
    # every rank reconstructs the same global list, rank-major
    assert all_count.tolist() == [2, 0, 5, 3, 0, 5]
    assert out[1] is None and out[4] is None and [len(o) for o in out if o is not None] == [2, 5, 3, 5]
    mine = out[rank * bs_local]
    assert torch.equal(mine, dets[0, :2 + rank])
    other = 1 - rank
    g2 = torch.Generator().manual_seed(100 + other)
    assert torch.equal(out[other * bs_local + 2], torch.rand(bs_local, cap, 7, generator=g2)[2, :5])
    # overflow of the gather capacity is an error, never a silent truncation
    try:
        split_gathered(*gather_detections(dets, torch.tensor([9, 0, 0], dtype=torch.int32), gather_cap=8))
        ok = False
    except RuntimeError:
        ok = True
    assert ok
    lo, hi = shard_bounds(7, world, rank)
    ret[rank] = (lo, hi)
    # uneven shards (7 images over 2 ranks: 4 + 3): padded on the wire, stripped afterwards, global order kept
    n_local = hi - lo
    d2 = torch.full((n_local, cap, 7), float(rank + 1))
    c2 = torch.arange(lo, hi, dtype=torch.int32) % 3
    a2, k2 = gather_detections(d2, c2, gather_cap=8)
    assert a2.shape == (7, 8, 7) and k2.tolist() == [i % 3 for i in range(7)]
    assert torch.all(a2[:4] == 1.0) and torch.all(a2[4:] == 2.0)
    # the same with the shard sizes handed in (shard_bounds: no size exchange on the wire), and with equal shards declared
    sizes = [b - a for a, b in (shard_bounds(7, world, r) for r in range(world))]
    a3, k3 = gather_detections(d2, c2, gather_cap=8, sizes=sizes)
    assert torch.equal(a3, a2) and torch.equal(k3, k2)
    a4, k4 = gather_detections(dets, count, gather_cap=8, equal_shards=True)
    assert torch.equal(a4, all_dets) and torch.equal(k4, all_count)
    try:
        gather_detections(d2, c2, gather_cap=8, sizes=[1, 1])      # sizes that contradict this rank's shard are an error
        ok = False
    except RuntimeError:
        ok = True
    assert ok
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_world2():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert dict(ret) == {0: (0, 4), 1: (4, 7)}


def _run_bench(*argv, env=None):
    import subprocess
    e = dict(os.environ)
    e.pop("RANK", None), e.pop("WORLD_SIZE", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e, timeout=300)


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` (no torch.distributed.run around it) starts two rank processes; rehearsed here without a
    GPU (--dry-run: gloo, real gather code, nothing measured).  The JSON line reports the ranks that really ran."""
    import json
    r = _run_bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and out["config"]["gather_ok"] is True
    assert out["value"] is None and "dry-run" in out["data"]


def test_bench_never_reports_fewer_ranks_than_asked():
    """No GPU here: `--gpus 2` must fail loudly (non-zero, no JSON line), and a WORLD_SIZE that disagrees with --gpus too."""
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "refusing" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())
    r = _run_bench("--gpus", "8", "--steps", "1", "--warmup", "0", env=dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
