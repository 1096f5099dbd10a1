"""Fixed-seed slices of the randomized sweeps (tools/fuzz_iir.py, tools/fuzz_fir.py) under -m gpu: random designs of
every IIR order 1..16 and random FIR tap counts, random channel counts and sequences of whole / ragged block lengths
with the state carried between them, through the C ABI against the oracle.  AUTO is what guards a user's design:
the conditioning probe must keep every draw within 1e-12 of the oracle relative to the output peak (or send it to
the bit-exact kernel); SEQUENTIAL must be bit-exact."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-12
BLOCKS = [1, 7, 100, 2048, 4096, 4096 + 5, 6144, 8192, 12288, 12288 + 2047, 16384]


@pytest.fixture(scope="module")
def G():
    from tests import gpu_util
    gpu_util.torch_mod()
    return gpu_util


def _iir_draw(rng, order):
    from dsp_amd import dsp
    if order % 2 == 0 and rng.random() < 0.5:
        lo = 0.15 + 0.2 * rng.random()
        f = dsp.butterworth(order // 2, lo, lo + 0.2 + 0.2 * rng.random())
    else:
        f = dsp.butterworth(order, 0.0, 0.2 + 0.5 * rng.random())
    b, a = np.array(f.b()), np.array(f.a())
    if rng.random() < 0.3:
        k = float(rng.choice([2.0, 0.5, -1.5]))
        b, a = k * b, k * a
    return b, a


@pytest.mark.parametrize("order", list(range(1, 17)))
def test_random_iir_designs_under_auto_and_sequential(G, order):
    from dsp_amd import bank as gbank
    rng = np.random.default_rng(1000 + order)
    for draw in range(3):
        b, a = _iir_draw(rng, order)
        nch = int(rng.choice([1, 3, 64, 130]))
        blocks = [int(rng.choice(BLOCKS)) for _ in range(int(rng.integers(2, 5)))]
        d = gbank.iir_design(b, a)
        ys_g, ys_o, gb, ob = G.run_pair([d], [0] * nch, blocks, gbank.MODE_AUTO, seed=100 * order + draw)
        err = float(np.max(util.rel_to_peak(np.concatenate(ys_g, axis=1), np.concatenate(ys_o, axis=1))))
        st, fv, ini = gb.get_state()
        serr = float(np.max(np.abs(st[:, :ob.state.shape[1]] - ob.state)) / max(1.0, float(np.max(np.abs(ob.state)))))
        assert err <= TOL and serr <= TOL, (order, draw, a[0], nch, blocks, err, serr)
        gb.close()
        if draw == 0:
            ys_g, ys_o, gb, ob = G.run_pair([d], [0] * nch, blocks, gbank.MODE_SEQUENTIAL, seed=100 * order + draw)
            assert util.bits_equal(np.concatenate(ys_g, axis=1), np.concatenate(ys_o, axis=1)), (order, blocks)
            G.assert_state_equal(gb, ob)
            gb.close()


@pytest.mark.parametrize("taps", [1, 2, 3, 8, 16, 17, 31, 32, 33, 64, 100, 255, 512, 700])
def test_random_fir_filters(G, taps):
    from dsp_amd import bank as gbank
    rng = np.random.default_rng(3000 + taps)
    for draw, mode in enumerate((gbank.MODE_AUTO, gbank.MODE_SEQUENTIAL)):
        b = rng.standard_normal(taps) / max(1, taps) ** 0.5
        partial = bool(rng.random() < 0.4)
        nch = int(rng.choice([1, 3, 64, 130]))
        blocks = [int(rng.choice([1, 7, 100, 2048, 4096, 4096 + 5, 8192, 12288 + 2047])) for _ in range(int(rng.integers(2, 4)))]
        d = gbank.fir_design(b, partial=partial)
        ys_g, ys_o, gb, ob = G.run_pair([d], [0] * nch, blocks, mode, seed=300 + taps + draw)
        yg, yo = np.concatenate(ys_g, axis=1), np.concatenate(ys_o, axis=1)
        if mode == gbank.MODE_SEQUENTIAL or taps < 2:          # AUTO: matrix-core kernel from 2 taps up
            assert util.bits_equal(yg, yo), (taps, partial, nch, blocks)
        else:
            assert float(np.max(util.rel_to_peak(yg, yo))) <= TOL, (taps, partial, nch, blocks)
        st, fv, ini = gb.get_state()
        assert util.bits_equal(st[:, :ob.state.shape[1]], ob.state) and np.array_equal(ini, ob.init)   # delay line exact
        gb.close()


def test_auto_bound_with_a_dc_offset(G):
    """What AUTO promises when the input is dominated by content the design rejects (include/RTFilterBank.h): a
    band-pass fed 1e5 + unit noise.  The state and the transient sit at the input's scale; the re-association error
    of the scan kernel stays within 1e-12 of THAT scale (and of the output's early peak), SEQUENTIAL is bit-exact."""
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    nch, n = 4, 1 << 15
    x = G.make_inputs(17, nch, n) + 1e5
    d = gbank.iir_design(b, a)
    ob = orc.Bank([G.to_orc_design(d)], [0] * nch)
    yo = ob.updaten(x, threads=4)
    gb = gbank.FilterBank([d], [0] * nch, mode=gbank.MODE_AUTO)
    y = gb(G.to_dev(x)).cpu().numpy()
    err = np.max(np.abs(y - yo), axis=1)
    assert np.all(err <= TOL * np.max(np.abs(x))), err
    assert np.all(err <= TOL * np.max(np.abs(yo), axis=1)), (err, np.max(np.abs(yo), axis=1))
    # late in the stream the output is unit-scale noise response while the error floor is set by the 1e5 offset
    tail = np.max(np.abs(y[:, n // 2:] - yo[:, n // 2:]))
    assert tail <= TOL * 1e5
    gs = gbank.FilterBank([d], [0] * nch, mode=gbank.MODE_SEQUENTIAL)
    assert util.bits_equal(gs(G.to_dev(x)).cpu().numpy(), yo)
    gb.close()
    gs.close()
