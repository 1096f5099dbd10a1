"""The two-wave pipelined kernel for orders 9..16 (dsp_amd/csrc/pipe16_kernel.hip), an opt-in alternative to
iir_scan16_kernel, against the oracle: as a stream whose segments are dealt over a grid of workgroups
(include/RTFilterStream.h: RTFilterStream_set_interleave), state carried over calls, fresh streams, in place,
ragged tails, every grid size from one workgroup to a full round.  1e-12 relative to the output peak."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def G():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests import gpu_util
    return gpu_util


@pytest.mark.parametrize("key", ["bw8_bp_0.25_0.75", "bw16_lp_0.5"])
@pytest.mark.parametrize("grid", [1, 3, 64, 1024])
def test_interleaved_stream_matches_oracle(G, key, grid):
    import torch
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    _, b, a = util.design_coefs(key)
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    if not fs.interleaved:
        pytest.skip("decay length of %s above one segment: %d" % (key, fs.decay_length))
    fs.set_interleave(grid)
    ref = orc.iir(b, a)
    pos = 0
    for n in (200000, 2048 * 64, 5, 1 << 20, 2048 * 700 + 13):      # the first call starts a fresh stream
        x = util.synth(11, 3, pos, n)
        xd = torch.from_numpy(x).cuda()
        yd = fs(xd)
        torch.cuda.synchronize()
        want = ref.updaten(x)
        err = np.max(np.abs(yd.cpu().numpy() - want)) / np.max(np.abs(want))
        assert err <= TOL, (key, grid, n, err)
        if n >= 2048 * 64:
            assert fs.last_split == min(grid, n // 2048)           # really went through the interleaved kernel
        pos += n
    st, fv, ini = fs.get_state()
    assert np.max(np.abs(st[:len(ref.state)] - ref.state)) <= TOL * max(1.0, np.max(np.abs(ref.state)))
    assert abs(fv - ref.filtered_value) <= TOL * max(1.0, abs(ref.filtered_value))
    assert ini == ref.initialized == 0
    fs.close()


def test_interleaved_stream_in_place_and_repeatable(G):
    """In place; the same call on the same data 20 times gives the same bits (the workgroups' hand-over of start
    states has no order-dependent arithmetic) -- the test that caught a store hazard in the publishing code."""
    import torch
    from dsp_amd import bank as gbank
    from dsp_amd import stream as gstream
    from oracle import oracle as orc
    _, b, a = util.design_coefs("bw8_bp_0.25_0.75")
    n = 2048 * 1024
    x = util.synth(5, 0, 0, n)
    want = orc.iir(b, a).updaten(x)
    first = None
    for rep in range(20):
        fs = gstream.FilterStream(gbank.iir_design(b, a))
        fs.set_interleave(1024)
        xd = torch.from_numpy(x).cuda()
        fs(xd, out=xd)
        torch.cuda.synchronize()
        got = xd.cpu().numpy()
        assert np.max(np.abs(got - want)) <= TOL * np.max(np.abs(want)), rep
        if first is None:
            first = got
        assert util.bits_equal(got, first), rep
        fs.close()


_BANK_SCRIPT = r"""
import numpy as np, torch
from dsp_amd import bank as gbank
from oracle import oracle as orc
from tests import util, gpu_util as G
worst = 0.0
for key, nch in (("bw8_bp_0.25_0.75", 70), ("bw16_lp_0.5", 9)):
    d = G.golden_design(key)
    bank = gbank.FilterBank([d], [0] * nch)
    ob = orc.Bank([G.to_orc_design(d)], [0] * nch)
    pos = 0
    for n in (2048 * 3, 2048, 2048 * 9, 1000, 2048 * 2):          # whole segments (pipelined kernel) and a ragged call
        x = G.make_inputs(3, nch, n, pos)
        y = bank(G.to_dev(x)).cpu().numpy()
        want = ob.updaten(x, threads=4)
        worst = max(worst, float(util.rel_to_peak(y, want).max()))
        pos += n
    st, fv, ini = bank.get_state()
    w = ob.state.shape[1]
    worst = max(worst, float(np.max(np.abs(np.asarray(st)[:, :w] - ob.state)) / max(1.0, np.max(np.abs(ob.state)))))
    worst = max(worst, float(np.max(np.abs(np.asarray(fv) - ob.fval)) / max(1.0, np.max(np.abs(ob.fval)))))
    assert np.all(np.asarray(ini) == 0)
    bank.close()
print("WORST %.3e" % worst)
"""


def test_bank_mode_in_a_process_of_its_own(G):
    """DSP_AMD_PIPE16=1 sends banks of order 9..16 to the pipelined kernel on calls of whole segments; the switch is
    read once per process, hence the child."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, DSP_AMD_PIPE16="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _BANK_SCRIPT], cwd=root, env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    worst = float(out.stdout.strip().split("WORST")[-1])
    assert worst <= TOL, worst
