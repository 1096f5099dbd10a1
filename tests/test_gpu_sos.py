"""GPU tests of second-order-section execution (RTFB_KIND_SOS / FILTER_SOS) through the
C ABI.  The reference never executes sections (its FILTER_SOS flag is not read), so the bar is the definition in
include/RTFilterDesign.h: every section is an RTIIRFilter (nb = na = 3) on the output of the one before it --
oracle.Cascade, a chain of reference-pinned oracle filters.  Sequential kernels: bit-exact against that chain;
scan kernels: <= 1e-12 of the output peak against it and against its 80-bit run."""
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


def _designs():
    from dsp_amd import bank as gbank
    from dsp_amd import dsp
    secs = [dsp.design_sos("butterworth", 4, 0.3, 0.6),        # 4 sections
            dsp.design_sos("butterworth", 2, 0.038, 0.042),    # 2 sections, the demo's filter
            dsp.design_sos("butterworth", 5, 0.0, 0.25),       # 3 sections, the last first-order
            dsp.design_sos("chebyshev2", 5, 0.4, 0.2, 0.1)]    # 5 sections, band-stop
    return secs, [gbank.sos_design(s) for s in secs]


@pytest.mark.parametrize("mode_name", ["sequential", "auto"])
@pytest.mark.parametrize("blocks", [[4096, 2048], [1, 1, 700, 5000, 3], [10000]])
def test_sos_bank_against_the_chain_of_reference_biquads(G, mode_name, blocks):
    from dsp_amd import bank as gbank
    from oracle import oracle as orc
    secs, designs = _designs()
    kind, b, a = util.design_coefs("bw4_bp_0.3_0.6")
    designs = designs + [gbank.iir_design(b, a), G.golden_design("fir32_lp_0.25")]       # mixed with TF IIR and FIR
    idx = [0, 1, 2, 3, 4, 5, 1, 0, 3, 2, 4, 1]
    mode = gbank.MODE_SEQUENTIAL if mode_name == "sequential" else gbank.MODE_AUTO
    gb = gbank.FilterBank(designs, idx, mode=mode)
    chains = [orc.Cascade(secs[d]) if d < 4 else (orc.iir(b, a) if d == 4 else orc.fir(designs[5]["b"])) for d in idx]
    pos = 0
    for n in blocks:
        x = G.make_inputs(31, len(idx), n, pos) * 100.0 + 3.0
        y = gb(G.to_dev(x)).cpu().numpy()
        for c, ch in enumerate(chains):
            yo = ch.updaten(x[c])
            if mode == gbank.MODE_SEQUENTIAL or idx[c] == 5 and len(designs[5]["b"]) < 17:
                assert util.bits_equal(y[c], yo), (c, idx[c], n)
            else:
                err = np.max(np.abs(y[c] - yo)) / max(np.max(np.abs(yo)), 1e-300)
                assert err <= TOL, (c, idx[c], n, err)
        pos += n
    st, fv, ini = gb.get_state()
    for c, ch in enumerate(chains):
        if idx[c] >= 4:
            continue
        ref = ch.state
        if mode == gbank.MODE_SEQUENTIAL:
            assert util.bits_equal(st[c, :len(ref)], ref) and fv[c] == ch.filtered_value
        else:
            assert np.max(np.abs(st[c, :len(ref)] - ref)) <= TOL * max(1.0, np.max(np.abs(ref)))
        assert not np.any(st[c, len(ref):]) and ini[c] == 0
    gb.close()


def test_demo_filter_reaches_the_scan_kernels_as_sections(G):
    """butterworth(2, 0.038, 0.042): as a transfer function the conditioning probe keeps it on the sequential
    kernel (scan error 5e-8); as two biquads it runs on the time-parallel kernels within 1e-12 of the 80-bit run
    of the same cascade."""
    import ctypes as C
    from dsp_amd import bank as gbank
    from dsp_amd import cdsp, dsp
    from oracle import oracle as orc
    f = dsp.butterworth(2, 0.038, 0.042)
    tf = gbank.iir_design(np.array(f.b()), np.array(f.a()))
    d = cdsp.RTFilterDesign_(tf["kind"], 5, 5, tf["b"].ctypes.data_as(C.POINTER(C.c_double)),
                             tf["a"].ctypes.data_as(C.POINTER(C.c_double)))
    err = C.c_double()
    assert cdsp.dspdll.dsp_scan_probe(C.byref(d), C.byref(err)) == 0 and err.value > 1e-13
    secs = dsp.design_sos("butterworth", 2, 0.038, 0.042)
    for sb, sa in secs:
        bb, aa = np.array(sb), np.array(sa)
        d = cdsp.RTFilterDesign_(0, 3, 3, bb.ctypes.data_as(C.POINTER(C.c_double)), aa.ctypes.data_as(C.POINTER(C.c_double)))
        assert cdsp.dspdll.dsp_scan_probe(C.byref(d), C.byref(err)) == 0 and err.value <= 1e-13, err.value
    nch, n = 8, 1 << 16
    x = G.make_inputs(5, nch, n)
    gb = gbank.FilterBank([gbank.sos_design(secs)], [0] * nch, mode=gbank.MODE_AUTO)
    y = gb(G.to_dev(x)).cpu().numpy()
    for c in range(nch):
        t = x[c]
        for sb, sa in secs:
            t = orc.iir_truth_ld(np.array(sb), np.array(sa), t)
        assert np.max(np.abs(y[c] - t)) / np.max(np.abs(t)) <= TOL
    gb.close()


def test_bank_from_host_sos_filters_and_sync_back(G):
    from dsp_amd import bank as gbank
    from dsp_amd import dsp
    hosts = [dsp.butterworth(4, 0.3, 0.6, flags=dsp.FILTER_SOS), dsp.butterworth(5, 0.0, 0.25, flags=dsp.FILTER_SOS),
             dsp.butterworth(4, 0.3, 0.6)]
    twins = [dsp.butterworth(4, 0.3, 0.6, flags=dsp.FILTER_SOS), dsp.butterworth(5, 0.0, 0.25, flags=dsp.FILTER_SOS),
             dsp.butterworth(4, 0.3, 0.6)]
    x = G.make_inputs(8, 3, 3000)
    head = [h(list(x[c, :100])) for c, h in enumerate(hosts)]          # the filters are already running
    for c, t in enumerate(twins):
        t(list(x[c, :100]))
    gb = gbank.FilterBank.from_filters(hosts, mode=gbank.MODE_SEQUENTIAL)
    y = gb(G.to_dev(x[:, 100:])).cpu().numpy()
    gb.sync_to_host()
    for c, t in enumerate(twins):
        yo = np.array(t(list(x[c, 100:])))
        assert util.bits_equal(y[c], yo)
        assert util.bits_equal(np.array(hosts[c].state()), np.array(t.state()))
        assert hosts[c].filtered_value == t.filtered_value and hosts[c].initialized == t.initialized
    assert head is not None
    gb.close()
