"""The pipelined IIR kernel on LOW orders, repeated: orders 1 and 2 have the shortest steps, so their segment
prefetch is the most likely to be still in flight when a segment is parked.  A build whose compiler had copied
prefetch registers in front of the hand-counted waits (pipe_kernel.hip: fetch; tools/check_prefetch_regs.py) ran the
last segment of a launch on stale samples in a few launches per hundred at exactly these shapes -- and passed every
single-shot parity test.  Every launch must reproduce the first one bit for bit, and the first must agree with the
bit-exact kernel to 1e-12."""
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


def _resonator(radius=0.9999, theta=0.3 * np.pi):
    b = np.array([(1.0 - radius) * 0.5, 0.0, -(1.0 - radius) * 0.5])
    a = np.array([1.0, -2.0 * radius * np.cos(theta), radius * radius])
    return b, a


@pytest.mark.parametrize("name", ["resonator2", "onepole1", "bw4_lp_0.5", "bw4_bp_0.3_0.6"])
@pytest.mark.parametrize("nch,n", [(256, 16384), (255, 16384), (1024, 2048), (700, 4096)])
def test_every_launch_reproduces_the_first(G, name, nch, n):
    import torch
    from dsp_amd import bank as gbank
    if name == "resonator2":
        b, a = _resonator()
    elif name == "onepole1":
        b, a = np.array([0.05, 0.05]), np.array([1.0, -0.9])
    else:
        _, b, a = util.design_coefs(name)
    x = torch.from_numpy(util.synth(77, 0, 0, nch * n).reshape(nch, n)).cuda()
    zero = (np.zeros((nch, len(a))), np.zeros(nch), np.zeros(nch, dtype=np.int32))
    exact = gbank.FilterBank([gbank.iir_design(b, a)], [0] * nch, mode=gbank.MODE_SEQUENTIAL)
    exact.set_state(*zero)
    want = exact(x)
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * nch)
    outs = [torch.empty_like(x) for _ in range(4)]
    first = None
    differ = 0
    for rep in range(60):
        bank.set_state(*zero)
        y = outs[rep % 4]
        bank(x, out=y)
        if first is None:
            torch.cuda.synchronize()
            first = y.clone()
            assert float((first - want).abs().max() / want.abs().max()) <= TOL
        elif not torch.equal(y, first):
            differ += 1
    torch.cuda.synchronize()
    assert differ == 0, "%d of 59 launches differ from the first" % differ
    bank.close()
    exact.close()
