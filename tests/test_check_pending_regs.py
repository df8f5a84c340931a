"""The assembly checker of the hand-scheduled kernels (tools/check_pending_regs.py) on two small listings of its own:
one that touches a register while its load is in flight, in the loop and across the loop edge, and one that waits
first.  A checker that cannot fail would prove nothing about the kernels it is run on."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_pending_regs  # noqa: E402

GOOD = """
demo_kernel_good: ; @demo_kernel_good
	s_load_dwordx2 s[2:3], s[0:1], 0x0
	v_mov_b32_e32 v9, 0
.LBB0_1:
	global_load_dwordx4 v[0:3], v8, s[2:3] offset:0
	global_load_dwordx4 v[4:7], v8, s[2:3] offset:16
	v_add_f32_e32 v9, v9, v9
	s_waitcnt vmcnt(1)
	v_add_f32_e32 v9, v0, v1
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v9, v4, v9
	global_store_dword v8, v9, s[2:3]
	s_cbranch_scc1 .LBB0_1
	s_endpgm
"""

BAD_IN_LOOP = GOOD.replace("demo_kernel_good", "demo_kernel_bad").replace(
    "\tv_add_f32_e32 v9, v9, v9\n", "\tv_mov_b32_e32 v10, v5\n")

# the load is issued at the bottom of the loop and waited for at the top of the next iteration, but a copy of its
# register sits between the issue and the branch (what the compiler does for a value that lives across the edge)
BAD_ACROSS_EDGE = """
demo_kernel_edge: ; @demo_kernel_edge
	global_load_dwordx4 v[0:3], v8, s[2:3] offset:0
.LBB1_1:
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v9, v0, v1
	global_load_dwordx4 v[0:3], v8, s[2:3] offset:0
	v_mov_b32_e32 v12, v2
	s_cbranch_scc1 .LBB1_1
	s_endpgm
"""


def _run(tmp_path, text, frag):
    f = tmp_path / "k.s"
    f.write_text(text)
    return check_pending_regs.check_file(str(f), [frag])


def test_clean_listing_has_no_violation(tmp_path):
    assert _run(tmp_path, GOOD, "demo_kernel_good") == (0, 1)


def test_touching_a_register_in_flight_is_found(tmp_path):
    bad, nfun = _run(tmp_path, BAD_IN_LOOP, "demo_kernel_bad")
    assert nfun == 1 and bad >= 1


def test_copy_across_the_loop_edge_is_found(tmp_path):
    bad, nfun = _run(tmp_path, BAD_ACROSS_EDGE, "demo_kernel_edge")
    assert nfun == 1 and bad >= 1


# spill traffic (scratch_*) counts against the same counter as the hand-issued loads and is walked as an entry: here
# the reloaded register is read while the reload is still in flight (vmcnt(1) retires the two global loads only)
SPILL_IN_FLIGHT = GOOD.replace("demo_kernel_good", "demo_kernel_spill").replace(
    "\tv_add_f32_e32 v9, v9, v9\n", "\tscratch_load_dword v11, off, s32 offset:4\n").replace(
    "\tv_add_f32_e32 v9, v0, v1\n", "\tv_add_f32_e32 v9, v0, v11\n")


def test_spill_reload_is_walked_as_an_entry(tmp_path):
    bad, nfun = _run(tmp_path, SPILL_IN_FLIGHT, "demo_kernel_spill")
    assert nfun == 1 and bad >= 1
    assert check_pending_regs.scratch_ops(str(tmp_path / "k.s"), ["demo_kernel_spill"]) == 1
    assert check_pending_regs.scratch_ops(str(tmp_path / "k.s"), ["no_such_kernel"]) == 0
