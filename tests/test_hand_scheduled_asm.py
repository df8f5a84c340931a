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


KERNEL_FILES = ["kernels_generic", "kernels_tiled2d", "kernels_persist2d", "kernels_persist_tti2d", "kernels_3d", "kernels_tti"]


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    """gfx950 assembly of every kernel file, compiled side by side (device code only)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm")
    csrc = os.path.join(ROOT, "devito-fwi_amd", "csrc")
    flags = subprocess.check_output(["make", "-s", "-C", csrc, "print-cxxflags"], text=True).split()
    jobs = {}
    for name in KERNEL_FILES:
        asm = str(out / (name + ".s"))
        jobs[name] = (asm, subprocess.Popen([HIPCC] + flags + ["--cuda-device-only", "-S", os.path.join(csrc, name + ".hip"),
                                                               "-o", asm], cwd=csrc, stderr=subprocess.DEVNULL))
    paths = {}
    for name, (asm, proc) in jobs.items():
        assert proc.wait() == 0, name
        paths[name] = asm
    return paths


def test_no_instruction_touches_a_register_with_a_load_in_flight(assembly):
    import check_pending_regs
    asm = assembly["kernels_tti"]
    KERNELS = ["tti3d_fused_kernel", "tti3d_step_kernel", "tti3d_step_wide_kernel", "tti3d_gz_kernel"]
    bad, nfun = check_pending_regs.check_file(asm, KERNELS)
    assert nfun == 10     # radius 2, 4 of the single march; 2, 4 and 6, 8 (wide) of the step, p = 1 .. 4 of the Gz march
    assert bad == 0
    # a spill reload between a hand-issued batch and its counted wait would make the wait one entry short: these
    # kernels must not spill at all
    assert check_pending_regs.scratch_ops(asm, KERNELS) == 0


@pytest.mark.parametrize("name", KERNEL_FILES)
def test_data_registers_of_wide_stores_are_left_alone(assembly, name):
    """A vector instruction that overwrites a data register of a 16-byte store in the issue slots right behind it can
    reach the store on gfx950 (run-to-run different results in the whole-loop TTI kernel once two tiles shared a CU;
    the compiler guards the constant-offset form only).  Every kernel of the library must leave such registers alone
    for at least 4 wait states (twice what the compiler inserts for the form it does guard); the kernel where the
    compiler left none is kernels_persist_tti2d.hip, whose stores go through store_f4_held (8 wait states)."""
    import check_store_data_hold
    asm = assembly[name]
    found = check_store_data_hold.check_file(asm, 4)
    assert not found, found[:4]


def test_store_hold_checker_sees_an_early_overwrite(tmp_path):
    import check_store_data_hold
    src = tmp_path / "t.s"
    src.write_text("k:\n\tbuffer_store_dwordx4 v[2:5], v35, s[52:55], s3 offen\n\tv_mov_b32_e32 v2, v11\n\ts_endpgm\n"
                   "k2:\n\tbuffer_store_dwordx4 v[2:5], v35, s[52:55], s3 offen\n\ts_nop 7\n\tv_mov_b32_e32 v2, v11\n\ts_endpgm\n"
                   "k3:\n\tglobal_store_dwordx4 v[0:1], v[6:9], off\n\tv_add_f32_e32 v0, v1, v2\n\tv_pk_mul_f32 v[8:9], v[0:1], v[2:3]\n\ts_endpgm\n")
    found = check_store_data_hold.check_file(str(src), 8)
    assert [(f, w) for f, _, _, w in found] == [("k", 0), ("k3", 1)]
