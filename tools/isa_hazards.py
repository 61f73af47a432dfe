#!/usr/bin/env python3
"""Static check of the hand-written MFMA kernels' ISA for the hazards hipcc cannot see through ``asm volatile("v_mfma...")``
(ADVICE r3): LLVM's hazard recognizer pads around the MFMAs it emits itself, an inline-asm MFMA is opaque to it.

Ground truth, measured on MI355X with tools/micro/mfma_war.hip (profiles/r04_mfma_hazard_probe.txt) for v_mfma_f32_16x16x32_bf16
(8 passes), idle and busy matrix pipe alike, distances in wait states (= issued instructions; ``s_nop N`` = N + 1):
  * a VALU WRITE of an A / B source register directly behind the MFMA (distance 0) does not change its result - the hazard that
    round 3's notes attributed to "a register an asm MFMA still reads" is not there;
  * a VALU READ of the D accumulator returns stale data at distances 0 - 6 and the result from 7 on (LLVM pads 8 + 2 = 10);
  * a VALU WRITE of D is overwritten by the MFMA's own write-back at distances 0 - 3 and survives from 4 on (LLVM pads 8 + 3 = 11).
So what an asm MFMA needs from the code around it is distance to every non-MFMA instruction that touches its D registers.

For every ``v_mfma`` of a kernel (device assembly from ``hipcc -S --cuda-device-only``), walking the following instructions in
program order (control flow is followed linearly, which covers the straight-line loop bodies these kernels consist of), ``analyse``
reports the smallest distance to
  * ``d_touch``: a non-MFMA instruction (VALU, LDS, vector memory) that reads or writes a D register - must be >= D_WINDOW (11);
  * ``d_reuse``: another MFMA that accumulates into the same D registers (back-to-back dependent MFMAs stall on the matrix pipe's own
    interlock; the kernels keep >= 25 others in between) - informational, pinned >= 4;
  * ``war_valu`` / ``war_async``: a VALU / an LDS or buffer load that writes an A or B source register - informational (see above).
Pinned by tests/test_host_cpu.py::test_asm_mfma_kernels_keep_their_accumulator_distance."""
import re
import subprocess
import sys

D_WINDOW = 11            # wait states LLVM keeps between an 8-pass MFMA and a VALU write of its D (passes + 3); measured need: 7 (read), 4 (write)
REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(op):
    out = set()
    for m in REG.finditer(op):
        lo = int(m.group(2) if m.group(2) is not None else m.group(3))
        hi = int(m.group(2) if m.group(2) is not None else m.group(4))
        out.update((m.group(1), r) for r in range(lo, hi + 1))
    return out


def parse(path):
    """{kernel name: [(mnemonic, [operand strings], line no)]}"""
    kernels, cur, name = {}, None, None
    for ln, line in enumerate(open(path), 1):
        s = line.split(";")[0].rstrip()
        if not s:
            continue
        m = re.match(r"^(_Z\w+):", s)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None or not s.startswith("\t") or s.lstrip().startswith("."):
            continue
        parts = s.strip().split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((parts[0], ops, ln))
    return kernels


def classify(mn, ops):
    """(kind, written registers, read registers); kind in mfma | valu | async | other"""
    if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
        return "mfma", regs(ops[0]), regs(ops[1]) | regs(ops[2]) | regs(ops[3])
    if mn.startswith("ds_read") or mn.startswith("ds_load"):
        return "async", regs(ops[0]), set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
    if re.match(r"(buffer|global|scratch|flat)_load", mn):
        if any("lds" in o.split() for o in ops):            # LDS-DMA: no vector destination
            return "other", set(), set().union(*[regs(o) for o in ops])
        return "async", regs(ops[0]), set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
    if re.match(r"(buffer|global|scratch|flat)_store|ds_write|ds_store", mn):
        return "other", set(), set().union(*[regs(o) for o in ops]) if ops else set()
    if mn.startswith("v_"):
        if mn.startswith("v_cmp") or mn.startswith("v_readlane") or mn.startswith("v_readfirstlane"):
            return "valu", set(), set().union(*[regs(o) for o in ops]) if ops else set()
        if mn.startswith("v_swap"):
            return "valu", regs(ops[0]) | regs(ops[1]), regs(ops[0]) | regs(ops[1])
        return "valu", regs(ops[0]), set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
    return "other", set(), set().union(*[regs(o) for o in ops]) if ops else set()


def analyse(insts, horizon=48):
    """Smallest wait-state distances behind any MFMA of the kernel: (VALU write of A/B, async write of A/B, non-MFMA touch of D)
    as (distance, line of the MFMA, line of the offender) or None."""
    cls = [classify(mn, ops) for mn, ops, _ in insts]
    best = {"war_valu": None, "war_async": None, "d_touch": None, "d_reuse": None}

    def note(key, dist, i, j):
        if best[key] is None or dist < best[key][0]:
            best[key] = (dist, insts[i][2], insts[j][2], insts[j][0])

    n_mfma = 0
    for i, (mn, ops, _) in enumerate(insts):
        if cls[i][0] != "mfma":
            continue
        n_mfma += 1
        d_regs = regs(ops[0])
        ab = regs(ops[1]) | regs(ops[2])
        ws = 0
        for j in range(i + 1, len(insts)):
            kind, wr, rd = cls[j]
            if kind == "valu" and wr & ab:
                note("war_valu", ws, i, j)
            if kind == "async" and wr & ab:
                note("war_async", ws, i, j)
            if kind != "mfma" and (wr | rd) & d_regs:
                note("d_touch", ws, i, j)
            if kind == "mfma" and wr & d_regs:
                note("d_reuse", ws, i, j)
            mnj = insts[j][0]
            ws += int(insts[j][1][0]) + 1 if mnj == "s_nop" else 1
            if ws > horizon:
                break
    return n_mfma, best


def device_asm(src, out, extra=()):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, "-S", "--cuda-device-only", "-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    return out


if __name__ == "__main__":
    for path in sys.argv[1:]:
        for name, insts in parse(path).items():
            n, best = analyse(insts)
            if n:
                print(f"{name[:90]}: {n} MFMAs")
                for k, v in best.items():
                    print(f"    {k:10s} {v}")
