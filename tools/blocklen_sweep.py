#!/usr/bin/env python3
"""Diagnostic: config-2 bank (1024 channels, order 8), kernel time vs block length: fixed cost per launch vs per-sample rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from tests import util

_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
C_, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 1 << 20
pitch = L + 128
xb = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
yb = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
gbank.synth_fill(xb[:, :L], 1)
bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_)
for n in (2048, 8192, 16384, 32768, 65536, 131072, 262144, 1 << 20):
    ts = []
    for rep in range(8):
        off = (rep * n) % (L - n + 1) if n < L else 0
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        bank.updaten_ptr(yb.data_ptr() + off * 8, pitch, xb.data_ptr() + off * 8, pitch, n)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = float(np.median(ts[2:]))
    print("%d ch x %7d: %.4f ms  %.0f GB/s" % (C_, n, t, C_ * n * 16 / t / 1e6), flush=True)
