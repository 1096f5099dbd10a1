"""Config 1, the literal drop-in: the REFERENCE's own python/cdsp.py + python/dsp.py, unmodified, loaded against
this repository's dsp.so, reproduce the demo's golden output bit for bit.  Runs only where /root/reference exists
(this container); nothing of the reference is copied -- a child interpreter imports it from where it lies.
The two accommodations SURVEY.md 8(c) lists are environmental, not source edits: `ctypes` pre-bound in builtins
(cdsp.py:15 uses the bare name) and the library directory on LD_LIBRARY_PATH."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PY = "/root/reference/python"

RUNNER = r"""
import builtins, ctypes, json, sys
builtins.ctypes = ctypes
sys.path.insert(0, sys.argv[1])
import numpy as np
import dsp                                     # the reference's module
x = np.load(sys.argv[2])["x"]
f = dsp.butterworth(2, 0.038, 0.042)
y_list = f(list(x))                            # Sequence -> RTFilter_updaten
g = dsp.butterworth(2, 0.038, 0.042)
y_scalar = [g(float(v)) for v in x[:64]]       # scalar -> RTFilter_update
gain, phase, freqs = dsp.filter_response_pzg([(0.5, 0.2), (0.5, -0.2)], [(1.0, 0.0)], kgain=0.5, n=16)
np.savez(sys.argv[3], y=np.array(y_list), ys=np.array(y_scalar), gain=np.array(gain), phase=np.array(phase))
"""


@pytest.mark.skipif(not os.path.isdir(REF_PY), reason="the reference tree is only present in the build container")
def test_reference_python_layer_runs_on_this_library(tmp_path):
    demo = util.load_npz("demo.npz")
    inp = tmp_path / "in.npz"
    out = tmp_path / "out.npz"
    np.savez(inp, x=demo["u"])
    env = dict(os.environ)
    libdir = os.path.join(ROOT, "dsp_amd", "lib")
    env["LD_LIBRARY_PATH"] = os.pathsep.join([libdir, "/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    # The reference tree ships a __pycache__ with byte code built by its authors; Python would load it instead of
    # compiling the .py sources.  A private cache prefix makes the child ignore that directory for reading and
    # writing: only source text of the reference is ever executed here.
    env["PYTHONPYCACHEPREFIX"] = str(tmp_path / "pycache")
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, "-c", RUNNER, REF_PY, str(inp), str(out)], env=env, cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    assert util.bits_equal(got["y"], demo["y"])
    assert util.bits_equal(got["ys"], demo["y"][:64])
    from dsp_amd import dsp as mine
    gain, phase, _ = mine.filter_response_pzg([(0.5, 0.2), (0.5, -0.2)], [(1.0, 0.0)], kgain=0.5, n=16)
    assert util.bits_equal(got["gain"], np.array(gain)) and util.bits_equal(got["phase"], np.array(phase))
