#!/usr/bin/env python3
"""The config-2 filter as four second-order sections (dsp.design_sos("butterworth", 4, 0.3, 0.6)), 1024 channels x
64K blocks: fused cascade (one pass) in AUTO and SEQUENTIAL mode, beside the transfer-function form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank, dsp
from tests import util
C, B, NB = 1024, 1 << 16, 8
x = torch.empty((NB, C, B), dtype=torch.float64, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
stream = torch.cuda.current_stream()
secs = dsp.design_sos("butterworth", 4, 0.3, 0.6)
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
def run(design, mode, nsteps=32):
    bank = gbank.FilterBank([design], [0] * C, device=0, mode=mode)
    for k in range(8):
        bank.updaten_ptr(y[k % NB].data_ptr(), B, x[k % NB].data_ptr(), B, B, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for k in range(nsteps):
        bank.updaten_ptr(y[k % NB].data_ptr(), B, x[k % NB].data_ptr(), B, B, stream)
    e1.record(stream)
    torch.cuda.synchronize()
    bank.close()
    us = e0.elapsed_time(e1) * 1e3 / nsteps
    return us, C * B / us
for name, d, mode in (("4 sections, AUTO (pipelined kernel, 8 states)", gbank.sos_design(secs), gbank.MODE_AUTO),
                      ("4 sections, SEQUENTIAL (lane per channel)", gbank.sos_design(secs), gbank.MODE_SEQUENTIAL),
                      ("transfer function, AUTO", gbank.iir_design(b, a), gbank.MODE_AUTO),
                      ("2 sections (demo filter), AUTO", gbank.sos_design(dsp.design_sos("butterworth", 2, 0.038, 0.042)), gbank.MODE_AUTO)):
    us, ms = run(d, mode)
    print("%-50s %8.1f us per 64K block = %7.0f Msamples/s = %.3f of 8 TB/s" % (name, us, ms, ms * 16 / 8e6), flush=True)
