#!/usr/bin/env python3
"""Order-16 bank on C channels x n samples for small C: with one workgroup per CU (C = 256) nothing is shared, and
time per 2048-sample segment per workgroup is the pipeline's own latency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from tests import util
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
s = torch.cuda.current_stream()
N = 1 << 18
for CH in (64, 256, 512, 768, 1024):
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * CH)
    x = torch.empty((CH, N), dtype=torch.float64, device="cuda")
    gbank.synth_fill(x, 1)
    y = torch.empty_like(x)
    for _ in range(20):
        bank.updaten_ptr(y.data_ptr(), N, x.data_ptr(), N, N, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(20):
        bank.updaten_ptr(y.data_ptr(), N, x.data_ptr(), N, N, s)
    e1.record(s)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%5d channels x 2^18: %8.1f us per launch, %.2f us per 2048-sample segment per workgroup, %.0f GB/s"
          % (CH, us, us / (N / 2048), CH * N * 16 / us / 1e3), flush=True)
    bank.close()
