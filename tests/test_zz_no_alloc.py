"""RTFilterBank_updaten_noalloc (include/RTFilterBank.h): the update entry point that allocates nothing.

Through the C ABI, against RTFilterBank_updaten on an identical bank: outputs and carried state to the bit
(same kernels, only the staging buffer of the in-place FIR call differs), on two calls in a row, the second
with a smaller input; the oracle checks the first call as well.  Collected last (file name) so that a failure
here cannot hide another test under `pytest -x`.
"""
import ctypes as C

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    from tests import gpu_util
    gpu_util.torch_mod()
    return gpu_util


def _banks(G, mode):
    from dsp_amd import bank as gbank
    designs = [G.golden_design("fir32_lp_0.25"), G.golden_design("bw4_bp_0.3_0.6"), G.golden_design("mavg32"),
               G.golden_design("fir512_lp_0.25")]
    idx = [0, 1, 2, 3, 0, 1, 2, 3, 3, 0]
    return designs, idx, gbank.FilterBank(designs, idx, mode=mode), gbank.FilterBank(designs, idx, mode=mode)


@pytest.mark.parametrize("mode_name", ["sequential", "auto"])
def test_noalloc_in_place_matches_updaten(G, mode_name):
    import torch
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    mode = gbank.MODE_SEQUENTIAL if mode_name == "sequential" else gbank.MODE_AUTO
    designs, idx, ref_bank, na_bank = _banks(G, mode)
    nch = len(idx)
    ob = orc.Bank([G.to_orc_design(d) for d in designs], idx)
    pos = 0
    for n in (6000, 1500):                       # two calls in a row, the second smaller
        x = G.make_inputs(77, nch, n, pos)
        buf_ref = G.to_dev(x)
        buf_na = G.to_dev(x)
        ref_bank.updaten_ptr(buf_ref.data_ptr(), n, buf_ref.data_ptr(), n, n)             # in place, may allocate
        used = gbank.dspdll.dsp_pool_bytes_in_use(0) + gbank.dspdll.dsp_pool_bytes_cached(0)
        na_bank.updaten_ptr_noalloc(buf_na.data_ptr(), n, buf_na.data_ptr(), n, n)        # in place, static scratch
        torch.cuda.synchronize()
        assert gbank.dspdll.dsp_pool_bytes_in_use(0) + gbank.dspdll.dsp_pool_bytes_cached(0) == used
        y_ref, y_na = buf_ref.cpu().numpy(), buf_na.cpu().numpy()
        assert util.bits_equal(y_na, y_ref), "noalloc entry point differs from RTFilterBank_updaten (n=%d)" % n
        yo = ob.updaten(x, threads=4)
        if mode == gbank.MODE_SEQUENTIAL:
            assert util.bits_equal(y_na, yo)
        else:
            assert float(np.max(util.rel_to_peak(y_na, yo))) <= 1e-12
        s_ref, s_na = ref_bank.get_state(), na_bank.get_state()
        for a, b in zip(s_ref, s_na):
            assert util.bits_equal(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))
        pos += n
    ref_bank.close()
    na_bank.close()


def test_noalloc_out_of_place_and_limit(G):
    import torch
    from dsp_amd import bank as gbank
    designs, idx, ref_bank, na_bank = _banks(G, gbank.MODE_AUTO)
    nch = len(idx)
    for n in (4096, 1000):
        x = G.to_dev(G.make_inputs(5, nch, n))
        y_ref, y_na = torch.empty_like(x), torch.empty_like(x)
        ref_bank.updaten_ptr(y_ref.data_ptr(), n, x.data_ptr(), n, n)
        na_bank.updaten_ptr_noalloc(y_na.data_ptr(), n, x.data_ptr(), n, n)
        torch.cuda.synchronize()
        assert util.bits_equal(y_na.cpu().numpy(), y_ref.cpu().numpy())
    # an in-place FIR call larger than the static array is refused with a size error, nothing is enqueued
    limit = gbank.noalloc_max_doubles()
    n_big = limit // nch + 8
    big = torch.zeros((nch, n_big), dtype=torch.float64, device="cuda")
    with pytest.raises(gbank.DSPError) as ei:
        na_bank.updaten_ptr_noalloc(big.data_ptr(), n_big, big.data_ptr(), n_big, n_big)
    assert ei.value.status == gbank.cdsp.RTFB_ESIZE
    # the same size out of place needs no staging and is accepted
    out = torch.empty_like(big)
    na_bank.updaten_ptr_noalloc(out.data_ptr(), n_big, big.data_ptr(), n_big, n_big)
    torch.cuda.synchronize()
    ref_bank.close()
    na_bank.close()


def test_noalloc_refusal_leaves_every_channel_untouched(G):
    """A refused call enqueues nothing: with the IIR channels FIRST in the bank (their group would run before the
    FIR group whose staging does not fit), the buffer and the carried state of every channel are what they were."""
    import torch
    from dsp_amd import bank as gbank
    designs = [G.golden_design("bw4_bp_0.3_0.6"), G.golden_design("fir32_lp_0.25")]
    idx = [0, 0, 0, 1, 1, 0]                           # an IIR channel at index 0
    bank = gbank.FilterBank(designs, idx, mode=gbank.MODE_AUTO)
    nch = len(idx)
    warm = G.to_dev(G.make_inputs(9, nch, 2048))
    bank.updaten_ptr_noalloc(warm.data_ptr(), 2048, warm.data_ptr(), 2048, 2048)      # a state worth comparing
    before = [np.array(a, copy=True) for a in bank.get_state()]
    n_big = gbank.noalloc_max_doubles() // 2 + 8       # the two FIR rows alone exceed the static array
    big = torch.rand((nch, n_big), dtype=torch.float64, device="cuda")
    keep = big.clone()
    with pytest.raises(gbank.DSPError) as ei:
        bank.updaten_ptr_noalloc(big.data_ptr(), n_big, big.data_ptr(), n_big, n_big)
    assert ei.value.status == gbank.cdsp.RTFB_ESIZE
    torch.cuda.synchronize()
    assert torch.equal(big, keep), "a refused call wrote into the caller's buffer"
    for a, b in zip(before, bank.get_state()):
        assert util.bits_equal(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))
    bank.close()
