#!/usr/bin/env python3
"""Config 2's tiled launches (1024 x 64K) by schedule period around the library's choice: median of several runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from tests import util
stream = torch.cuda.current_stream()
Cn, n = 1024, 1 << 16
x = torch.empty(16 * Cn * n, dtype=torch.float64, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
def run(pace, nsteps=64):
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * Cn, device=0)
    bank.set_pace(pace)
    def call(k):
        off = (k % 16) * Cn * n * 8
        bank.updaten_tiled_ptr(y.data_ptr() + off, x.data_ptr() + off, n, stream)
    for k in range(320): call(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for k in range(nsteps): call(k)
    e1.record(stream)
    torch.cuda.synchronize()
    bank.close()
    return Cn * n * nsteps / (e0.elapsed_time(e1) / 1e3) / 1e6
base = Cn * 32768.0 / 5950.0
fs = [0.94, 0.96, 0.97, 0.98, 0.99, 1.0, 1.01, 1.02]
res = {f: [] for f in fs}
for rep in range(5):
    for f in fs:
        res[f].append(run(base * f))
for f in fs:
    v = res[f]
    print("period x %.2f (%.0f ns): median %.0f  min %.0f  max %.0f Msamples/s" % (f, base * f, np.median(v), min(v), max(v)), flush=True)
