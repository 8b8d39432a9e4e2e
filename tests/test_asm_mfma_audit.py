"""kz_tower4.hip issues its MFMAs as asm statements (the only way to keep 256 accumulator registers pinned), and hipcc
pads no hazard around an asm statement.  The rule the kernel relies on — no compiler-generated accumulator read inside
the wait states of the MFMA that wrote it, no compiler write into the accumulator file among the MFMAs, no VALU-written
MFMA operand right in front of the statement, no scratch — is checked on the generated ISA (cross-compiled, no GPU)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.timeout(600)
def test_tower4_isa_has_no_unpadded_hazard_and_no_scratch(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    src = os.path.join(REPO, "experiments", "csrc", "kz_tower4.hip")  # (a rejected kernel: experiments/README.md)
    asm = str(tmp_path / "kz_tower4.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5",
                           "-Wno-unused-result", "-I" + os.path.join(REPO, "kzero_amd", "csrc"), "-S", "--cuda-device-only", "-o", asm, src],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "audit_asm_mfma.py"), asm], capture_output=True,
                         text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    text = open(asm).read()
    assert ".vgpr_spill_count: 0" in text or ".vgpr_spill_count: 6" in text  # (6: two cold values outside the loops)
    assert text.count("v_mfma_f32_16x16x32_f16") > 1400                      # the three tap lines + the stem
