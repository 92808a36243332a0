"""CPU: a check of the generated ISA (hipcc cross-compiles gfx950 without a GPU).

ADVICE r3 (chol.hip, p2_read / p2_wait): the diagonal-block kernel issues its LDS fragment reads from inline asm
(`ds_read_b64`) and waits for them with a hand-written `s_waitcnt lgkmcnt(0)` in a LATER asm statement; the compiler's
wait-count insertion does not track inline-asm results, so nothing but the register allocator's mercy keeps it from placing
a copy or a spill of those VGPRs between the read and the wait -- which would read stale data.  This test compiles
chol.hip (which includes the GEMM core's main loop, same pattern with COUNTED waits) to assembly and walks every kernel:
LDS reads return in order, so an inline-asm `s_waitcnt lgkmcnt(N)` leaves the N youngest reads pending; no
compiler-generated instruction may touch a register a still-pending read will write."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vregs(tok):
    """the VGPR numbers an operand token names: v12 -> {12}, v[4:7] -> {4..7}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_no_compiler_instruction_touches_a_pending_inline_asm_lds_read(tmp_path):
    out = tmp_path / "chol.s"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function",
                           "-S", "--cuda-device-only", os.path.join(ROOT, "madqp_jl_amd", "csrc", "chol.hip"), "-o", str(out)])
    lines = out.read_text().splitlines()
    kernel, in_asm, pending, checked, pairs = None, False, [], 0, 0
    for ln in lines:
        t = ln.strip()
        m = re.match(r"^(_ZN\S*):\s", ln)
        if m:
            kernel, pending = m.group(1), []
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        ops = [o.strip() for o in re.split(r"[,\s]+", t)[1:]]
        if in_asm:
            if t.startswith("ds_read_b64"):
                pending.append(vregs(ops[0]))
                pairs += 1
            elif t.startswith("s_waitcnt"):
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m:  # in-order return: only the N youngest reads can still be outstanding
                    n = int(m.group(1))
                    pending = pending[len(pending) - n:] if n else []
            continue
        if pending:  # a compiler-generated instruction while inline-asm reads are in flight
            checked += 1
            touched = set().union(*[vregs(o) for o in ops]) if ops else set()
            busy = set().union(*pending)
            assert not (touched & busy), f"{kernel}: `{t}` touches v{sorted(touched & busy)} before the s_waitcnt of its ds_read"
            m = re.search(r"lgkmcnt\((\d+)\)", t) if t.startswith("s_waitcnt") else None
            if m:  # (a compiler-inserted wait counts too)
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n else []
            if t.startswith("s_endpgm"):
                pending = []
    assert pairs > 100, "the inline-asm LDS reads of the diagonal-block kernel were not found: the pattern changed, adapt this test"
