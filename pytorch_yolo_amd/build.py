"""In-tree build of libyolo_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m pytorch_yolo_amd.build [--force]

The .so stays next to the sources (git-ignored, but it travels with gpurun snapshots).
"""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libyolo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# per-file extra flags: nms.hip must not contract a*b+c into fma (bit-exact IoU, see its header)
SOURCES = {
    "runtime.hip": [],
    "conv_igemm.hip": [],
    "conv3x3_halo.hip": [],
    "conv3x3_t20.hip": [],
    "conv_resunit.hip": [],
    "conv_resunit_t20.hip": [],
    "conv_stem.hip": [],
    "conv_mbconv.hip": [],
    "conv_mbwide.hip": [],
    "conv_small.hip": [],
    "conv_f32.hip": ["-ffp-contract=off"],
    "conv1x1_stream.hip": [],
    "pointwise.hip": [],
    "efficient.hip": [],
    "preprocess.hip": ["-ffp-contract=off"],
    "nms.hip": ["-ffp-contract=off"],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function"]


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(CSRC, "nms_common.h"),
               os.path.join(CSRC, "head_epilogue.h"),
               os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "yolo_hip.h"),
               os.path.abspath(__file__)]        # per-file flags live here: a flag change rebuilds too
    objs, jobs = [], []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC, *COMMON, *extra, "-c", s, "-o", o])
            for stale in (o, LIB_PATH):          # never leave an out-of-date library behind a failed build
                if os.path.exists(stale):
                    os.remove(stale)

    def compile_one(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    if jobs:        # one hipcc per source, a few at a time (YOLO_BUILD_JOBS; the files are independent)
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(len(jobs), int(os.environ.get("YOLO_BUILD_JOBS", str(min(6, os.cpu_count() or 1))))))
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(compile_one, jobs))
    if force or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
