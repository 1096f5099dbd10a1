"""Host-side streaming around the device path: RTFilterBank_updaten_host (the reference's
per-filter call on ordinary host arrays, staged through pinned rings) and RTFilterBankFeed (per-sample producers
coalesced into device blocks).  Checked against the oracle through the C ABI."""
import time

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def G():
    from tests import gpu_util
    gpu_util.torch_mod()
    return gpu_util


def _mixed(G):
    from dsp_amd import bank as gbank
    designs = [G.golden_design("bw4_bp_0.3_0.6"), G.golden_design("fir32_lp_0.25"), G.golden_design("mavg32"),
               G.golden_design("bw8_bp_0.25_0.75"), G.golden_design("fir7_lp_0.4")]
    return designs


def test_bank_updaten_host_matches_the_oracle(G):
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    designs = _mixed(G)
    nch = 40
    idx = [c % len(designs) for c in range(nch)]
    gb = gbank.FilterBank(designs, idx, mode=gbank.MODE_SEQUENTIAL)
    ob = orc.Bank([G.to_orc_design(d) for d in designs], idx)
    pos = 0
    for n in (450000, 1, 3000):                      # several staging chunks; a single sample; a short block
        x = G.make_inputs(55, nch, n, pos)
        y = gb.filter_host(x)
        assert util.bits_equal(y, ob.updaten(x, threads=8)), n
        pos += n
    # rows with a pitch, in place
    n = 250000
    buf = np.zeros((nch, n + 37))
    buf[:, :n] = G.make_inputs(55, nch, n, pos)
    want = ob.updaten(buf[:, :n], threads=8)
    gb.filter_host(buf[:, :n], out=buf[:, :n])
    assert util.bits_equal(buf[:, :n], want) and not np.any(buf[:, n:])
    G.assert_state_equal(gb, ob)
    gb.close()


def test_bank_updaten_host_throughput_is_reported(G):
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    _, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    nch, n = 1024, 1 << 17                            # 1 GiB each way
    gb = gbank.FilterBank([gbank.iir_design(b, a)], [0] * nch)
    x = np.random.default_rng(1).standard_normal((nch, n))
    y = np.empty_like(x)
    gb.filter_host(x[:, :4096], out=y[:, :4096])     # builds the rings
    gb.reset()
    t0 = time.perf_counter()
    gb.filter_host(x, out=y)
    dt = time.perf_counter() - t0
    print("\\nRTFilterBank_updaten_host: %d x 2^17 samples in %.3f s = %.0f Msamples/s (%.1f GB/s each way)"
          % (nch, dt, nch * n / dt / 1e6, nch * n * 8 / dt / 1e9))
    for c in (0, 511, 1023):
        yo = orc.iir(b, a).updaten(x[c])
        assert np.max(np.abs(y[c] - yo)) <= TOL * np.max(np.abs(yo))
    gb.close()


def test_feed_coalesces_per_sample_frames(G):
    """Frames pushed one by one and in bunches come back, in order, as RTFilter_update would have produced them
    channel by channel (bit-exact kernels), with the state carried from block to block."""
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    designs = _mixed(G)
    nch = 23
    idx = [c % len(designs) for c in range(nch)]
    gb = gbank.FilterBank(designs, idx, mode=gbank.MODE_SEQUENTIAL)
    ob = orc.Bank([G.to_orc_design(d) for d in designs], idx)
    total, block = 1700, 256
    x = G.make_inputs(9, nch, total)                 # (nch, total): frame t = x[:, t]
    want = ob.updaten(x, threads=4)
    feed = gbank.BankFeed(gb, block)
    got = []
    t = 0
    rng = np.random.default_rng(2)
    while t < total:
        k = int(rng.choice([1, 1, 1, 7, 100, 300]))
        k = min(k, total - t)
        chunk = np.ascontiguousarray(x[:, t:t + k].T)
        while len(chunk):
            took = feed.push(chunk)
            chunk = chunk[took:]
            if len(chunk):                                    # both output buffers hold unpopped frames
                got.append(feed.pop(4096, wait=True))
        t += k
        out = feed.pop(4096)                                  # poll: whatever is back
        if len(out):
            got.append(out)
    feed.flush()
    assert feed.available() == total - sum(len(o) for o in got)
    got.append(feed.pop(4096))
    y = np.concatenate(got).T
    assert y.shape == want.shape and util.bits_equal(y, want)
    # a third block before anything is popped is refused, not silently dropped
    assert feed.pending() == 0
    assert feed.push(np.zeros((2 * block + 5, nch))) == 2 * block and feed.pending() == 2
    assert gbank.dspdll.RTFilterBankFeed_push(feed._h, None, 0, None) == 0
    one = np.zeros(nch)
    import ctypes as C
    assert gbank.dspdll.RTFilterBankFeed_push(feed._h, one.ctypes.data_as(C.POINTER(C.c_double)), 1, None) \
        == gbank.cdsp.RTFB_ESIZE
    feed.close()
    gb.close()
