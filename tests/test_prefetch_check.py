"""tools/check_prefetch_regs.py on the ISA listings the build leaves in dsp_amd/csrc/build/ (the Makefile runs the same
check and fails the build on a finding; this is the record that it ran on the tree under test), and on two
hand-made listings, one clean and one with the copy it exists to catch."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_prefetch_regs.py")

_CLEAN = """
_ZN3dsp15iir_pipe_kernelILi2ELb1ELb1ELb0ELb0EEEvNS_9GroupViewENS_7BlockIOENS_7PipeExtE:
	global_load_dwordx4 v[62:65], v[66:67], off nt
	global_load_dwordx4 v[58:61], v[66:67], off offset:1024 nt
	;;#ASMSTART
	s_waitcnt vmcnt(17)
	;;#ASMEND
	ds_write_b128 v158, v[62:65]
	;;#ASMSTART
	s_waitcnt vmcnt(16)
	;;#ASMEND
	ds_write_b128 v166, v[58:61] offset:1024
	s_endpgm
"""

_COPIED = _CLEAN.replace("	;;#ASMSTART\n	s_waitcnt vmcnt(17)", "	v_mov_b64_e32 v[130:131], v[62:63]\n	v_mov_b64_e32 v[132:133], v[64:65]\n	;;#ASMSTART\n	s_waitcnt vmcnt(17)").replace(
    "ds_write_b128 v158, v[62:65]", "ds_write_b128 v158, v[130:133]")


def _run(paths):
    return subprocess.run([sys.executable, TOOL] + list(paths), capture_output=True, text=True)


def test_checker_tells_a_copy_from_a_clean_park(tmp_path):
    clean, copied = tmp_path / "clean.s", tmp_path / "copied.s"
    clean.write_text(_CLEAN)
    copied.write_text(_COPIED)
    assert _run([str(clean)]).returncode == 0
    out = _run([str(copied)])
    assert out.returncode == 1 and "COPIED PREFETCH REGISTERS" in out.stdout
    empty = tmp_path / "empty.s"
    empty.write_text("nothing here\n")
    assert _run([str(empty)]).returncode == 2


def test_the_built_kernels_park_from_their_load_registers():
    listings = sorted(glob.glob(os.path.join(ROOT, "dsp_amd", "csrc", "build", "pipe*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if not listings:
        pytest.skip("no ISA listings (the library was built elsewhere)")
    out = _run(listings)
    assert out.returncode == 0, out.stdout[-3000:]
    assert out.stdout.count("ok ") >= 40          # every whole-segment instantiation of both kernels was looked at
