#!/usr/bin/env python3
"""Diagnostic: lane-per-channel sequential IIR kernel (bit-exact) vs time-parallel scan kernels on wide banks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from tests import util

SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(65536, 16384), (262144, 4096), (16384, 65536)]
for key in ("bw4_bp_0.3_0.6", "bw8_bp_0.25_0.75", "bw5_lp_0.25"):
    _, b, a = util.design_coefs(key)
    for C_, B in SHAPES:
        pitch = B + 128
        xb = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
        yb = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
        gbank.synth_fill(xb[:, :B], 1)
        res = []
        for mode in (gbank.MODE_SEQUENTIAL, gbank.MODE_SCAN):
            bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C_, mode=mode)
            ts = []
            for rep in range(5):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                bank.updaten_ptr(yb.data_ptr(), pitch, xb.data_ptr(), pitch, B)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res.append(np.median(ts[2:]))
            bank.close()
        print("%-18s %7d ch x %6d: sequential (bit-exact) %.3f ms %.0f GB/s   scan %.3f ms %.0f GB/s" %
              (key, C_, B, res[0], C_ * B * 16 / res[0] / 1e6, res[1], C_ * B * 16 / res[1] / 1e6), flush=True)
        del xb, yb
