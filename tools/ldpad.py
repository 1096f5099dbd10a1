#!/usr/bin/env python3
"""Diagnostic: scan kernel time vs leading-dimension padding (HBM channel camping check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from tests import util

_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
C_, L, B = 1024, 1 << 19, 1 << 16
for pad in (0, 32, 96, 512, 544, 2080, 8224):
    ld = L + pad
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_, mode=gbank.MODE_SCAN)
    xb = torch.empty(C_ * ld, dtype=torch.float64, device="cuda")
    yb = torch.empty(C_ * ld, dtype=torch.float64, device="cuda")
    x = xb.view(C_, ld)[:, :L]
    gbank.synth_fill(x, 1)
    ts = []
    for rep in range(3):
        for j in range(L // B):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            bank.updaten_ptr(yb.data_ptr() + j * B * 8, ld, xb.data_ptr() + j * B * 8, ld, B)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    t = np.median(ts[8:])
    print("ld pad %6d doubles: kernel %.3f ms  %.1f GB/s (min %.3f ms)" % (pad, t, C_ * B * 16 / t / 1e6, min(ts[8:])))
    bank.close(); del xb, yb, x
