#!/usr/bin/env python3
"""Diagnostic: bit-exact VALU FIR kernel vs Toeplitz MFMA kernel by tap count (run with DSP_AMD_FIR_MFMA_MIN=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank

C_, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 1 << 16)
pitch = B + 128
xb = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
yb = torch.empty((C_, pitch), dtype=torch.float64, device="cuda")
gbank.synth_fill(xb[:, :B], 1)
rng = np.random.default_rng(0)
for K in ((2, 4, 8, 12, 16, 24, 32, 48, 64, 128) if len(sys.argv) <= 3 else [int(v) for v in sys.argv[3:]]):
    taps = rng.standard_normal(K) / K
    res = []
    for mode in (gbank.MODE_SEQUENTIAL, gbank.MODE_AUTO):
        bank = gbank.FilterBank([gbank.fir_design(taps)], [0] * C_, mode=mode)
        ts = []
        for rep in range(6):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            bank.updaten_ptr(yb.data_ptr(), pitch, xb.data_ptr(), pitch, B)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append(np.median(ts[2:]))
        bank.close()
    print("K=%4d  VALU (bit-exact) %.3f ms %.0f GB/s   MFMA %.3f ms %.0f GB/s" %
          (K, res[0], C_ * B * 16 / res[0] / 1e6, res[1], C_ * B * 16 / res[1] / 1e6), flush=True)
