#!/usr/bin/env python3
"""Diagnostic: bit-exact (RTFB_MODE_SEQUENTIAL) and AUTO FIR kernels by tap count: 16384 channels x 64K samples
(16 GiB moved per call), TB/s of algorithmic traffic.  usage: fir_sweep.py [taps ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
CH, N = 16384, 65536
taps = [int(v) for v in sys.argv[1:]] or [2, 4, 6, 8, 12, 16, 24, 32, 33, 48, 64, 96, 128]
x = torch.empty((CH, N), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
gbank.synth_fill(x, 3)
s = torch.cuda.current_stream()
rng = np.random.default_rng(0)
for K in taps:
    b = rng.standard_normal(K) / K
    row = "taps %4d :" % K
    for name, mode in (("bit-exact", gbank.MODE_SEQUENTIAL), ("auto", gbank.MODE_AUTO)):
        bank = gbank.FilterBank([gbank.fir_design(b)], [0] * CH, mode=mode)
        for _ in range(2):
            bank.updaten_ptr(y.data_ptr(), N, x.data_ptr(), N, N, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(4):
            bank.updaten_ptr(y.data_ptr(), N, x.data_ptr(), N, N, s)
        e1.record(s)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        row += "  %s %7.3f ms %5.2f TB/s" % (name, ms, CH * N * 16 / ms / 1e9)
        bank.close()
    print(row, flush=True)
