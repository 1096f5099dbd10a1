"""Helpers for the -m gpu tests: device buffers via torch, checker via oracle/."""
import numpy as np

from oracle import oracle as orc
from dsp_amd import bank as gbank
from tests import util


def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def to_dev(x):
    torch = torch_mod()
    return torch.from_numpy(np.ascontiguousarray(x)).to("cuda")


def to_orc_design(d):
    if d["kind"] == gbank.KIND_IIR:
        return dict(kind=orc.KIND_IIR, init_mode=orc.INIT_IIR_STABLE, b=d["b"], a=d["a"])
    mode = orc.INIT_FIR_PARTIAL if d["kind"] == gbank.KIND_FIR_PARTIAL else orc.INIT_FIR_STABLE
    return dict(kind=orc.KIND_FIR, init_mode=mode, b=d["b"])


def golden_design(name):
    kind, b, a = util.design_coefs(name)
    if kind == "iir":
        return gbank.iir_design(b, a)
    return gbank.fir_design(b, partial=(kind == "fir_partial"))


def make_inputs(seed, nch, n, n0=0):
    return np.stack([util.synth(seed, c, n0, n) for c in range(nch)])


def run_pair(designs, design_idx, blocks, mode, seed=1234, threads=8):
    """Run the same blocks through the GPU bank and the oracle bank.

    Returns (list of gpu outputs, list of oracle outputs, gpu bank, oracle bank)."""
    nch = len(design_idx)
    gb = gbank.FilterBank(designs, design_idx, mode=mode)
    ob = orc.Bank([to_orc_design(d) for d in designs], design_idx)
    ys_g, ys_o, pos = [], [], 0
    for n in blocks:
        x = make_inputs(seed, nch, n, pos)
        ys_g.append(gb(to_dev(x)).cpu().numpy())
        ys_o.append(ob.updaten(x, threads=threads))
        pos += n
    return ys_g, ys_o, gb, ob


def assert_state_equal(gb, ob, exact=True, tol=0.0):
    st, fv, ini = gb.get_state()
    w = ob.state.shape[1]
    if exact:
        assert util.bits_equal(st[:, :w], ob.state)
        assert util.bits_equal(fv, ob.fval)
    else:
        scale = max(1.0, float(np.max(np.abs(ob.state))))
        assert np.max(np.abs(st[:, :w] - ob.state)) <= tol * scale
        assert np.max(np.abs(fv - ob.fval)) <= tol * scale
    assert np.array_equal(ini, ob.init)
