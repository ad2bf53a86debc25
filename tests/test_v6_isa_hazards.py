"""awq_gemm_v6.hip issues its MFMAs as inline asm (accumulators pinned in AGPRs), so hipcc's hazard recogniser does not see them.  The
one hazard that bit (a VALU write directly in front of the MFMA that reads the register: the MFMA saw the stale value) is kept out by
construction -- every dequant MFMA statement carries its own s_nop, the product MFMAs' operands are written at least one MFMA slot
earlier -- and this test checks the GENERATED ISA of every instantiation for it: no v_* instruction may write a source register of
the MFMA that directly follows it with fewer than two wait states in between.  CPU-only (hipcc cross-compiles gfx950)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


# every source that issues MFMAs from inline asm: the product kernel and the AWQ_PROBES-only experiment on the planned 32-row interleave
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("source,defines,min_checked", [("awq_gemm_v6.hip", [], 1000)])
def test_no_valu_write_directly_in_front_of_an_asm_mfma_that_reads_it(source, defines, min_checked):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *defines, "-I", os.path.join(ROOT, "include"), "-I",
               os.path.join(ROOT, "llm_awq_amd", "csrc"), "-mllvm", "-amdgpu-kernarg-preload-count=16", "-S", "--cuda-device-only",
               os.path.join(ROOT, "llm_awq_amd", "csrc", source), "-o", out]
        subprocess.run(cmd, check=True, capture_output=True, timeout=600)
        lines = [ln.strip() for ln in open(out)]
    lines = [ln for ln in lines if ln and not ln.startswith(";") and not ln.startswith(".") and "ASMSTART" not in ln and "ASMEND" not in ln]
    checked, bad = 0, []
    for i, ln in enumerate(lines):
        if not (ln.startswith("v_mfma_f32_16x16x32") or ln.startswith("v_mfma_f32_32x32x16") or ln.startswith("v_mfma_f32_4x4x4")):
            continue
        checked += 1
        srcs = set()
        for tok in [t.strip() for t in ln.split(None, 1)[1].split(",")][1:]:
            srcs |= _regs(tok)
        j, wait = i - 1, 0
        while j >= 0 and lines[j].startswith("s_nop"):
            wait += int(lines[j].split()[1]) + 1
            j -= 1
        prev = lines[j] if j >= 0 else ""
        if prev.startswith("v_") and not prev.startswith("v_mfma") and wait < 2:
            if _regs(prev.split(None, 1)[1].split(",")[0].strip()) & srcs:
                bad.append((prev, ln))
    assert checked > min_checked, checked      # every instantiation's K loop was seen
    assert not bad, bad[:5]
