"""CPU oracle for the YOLOv3 inference hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, in plain fp32 torch / numpy
on the CPU, what the reference (Dipet/pytorch_yolo) computes on the path
forward() -> YOLOLayer decode -> non_max_suppression().  It exists so that the
HIP kernels can be checked against it.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``pytorch_yolo_amd`` never does, and never falls back
to it.

Pinning: the oracle is checked tensor-for-tensor against the reference itself
(imported read-only in the build container by ``tests/golden/make_golden.py``)
and against the golden vectors that script committed under ``tests/golden/``
(``tests/test_oracle_golden.py``).  One part is **parity unpinned**: the
MobileNetV2 encoder, whose arithmetic lives in torchvision (pinned 0.3.0 in
/root/reference/requirements.txt:8) which is absent here; ``oracle/mobilenet.py``
restates the published architecture and says so in its header.
"""
