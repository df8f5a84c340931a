"""The marching 3-D TTI kernels (csrc/kernels_tti.hip: tti3d_fused_kernel, tti3d_gz_kernel,
tti3d_step_kernel, tti3d_step_wide_kernel) issue their loads through
inline assembly and wait for them by hand; the compiler does not know that the destination registers are still being
written.  This test compiles the file to gfx950 assembly and walks the kernels: between a load and the wait that
covers it nothing may touch the load's registers (a register copy or spill placed there by the compiler would read
garbage on the GPU, intermittently).  No GPU needed."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_instruction_touches_a_register_with_a_load_in_flight(tmp_path):
    import check_pending_regs
    csrc = os.path.join(ROOT, "devito-fwi_amd", "csrc")
    flags = subprocess.check_output(["make", "-s", "-C", csrc, "print-cxxflags"], text=True).split()
    asm = str(tmp_path / "kernels_tti.s")
    subprocess.check_call([HIPCC] + flags + ["--cuda-device-only", "-S", os.path.join(csrc, "kernels_tti.hip"), "-o", asm],
                          cwd=csrc, stderr=subprocess.DEVNULL)
    KERNELS = ["tti3d_fused_kernel", "tti3d_step_kernel", "tti3d_step_wide_kernel", "tti3d_gz_kernel"]
    bad, nfun = check_pending_regs.check_file(asm, KERNELS)
    assert nfun == 10     # radius 2, 4 of the single march; 2, 4 and 6, 8 (wide) of the step, p = 1 .. 4 of the Gz march
    assert bad == 0
    # a spill reload between a hand-issued batch and its counted wait would make the wait one entry short: these
    # kernels must not spill at all
    assert check_pending_regs.scratch_ops(asm, KERNELS) == 0
