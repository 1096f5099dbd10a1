#!/usr/bin/env python3
"""Diagnostic: kernel time vs channel count for the scan kernel (residency / scaling check)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from tests import util

_, b, a = util.design_coefs(sys.argv[1] if len(sys.argv) > 1 else "bw4_bp_0.3_0.6")
L, B = 1 << 18, 1 << 16
for C_ in (256, 512, 768, 1024, 1280, 1536, 2048):
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_, mode=gbank.MODE_SCAN)
    x = torch.empty((C_, L), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    gbank.synth_fill(x, 1)
    ts = []
    for rep in range(3):
        for j in range(L // B):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            bank.updaten_ptr(y.data_ptr() + j * B * 8, L, x.data_ptr() + j * B * 8, L, B)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    t = np.median(ts[4:])
    print("channels %5d  kernel %.3f ms  %.1f GB/s  %.3f us/channel" % (C_, t, C_ * B * 16 / t / 1e6, t * 1e3 / C_))
    bank.close(); del x, y
