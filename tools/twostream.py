#!/usr/bin/env python3
"""Diagnostic: one 1024-channel bank on one stream vs k banks of 1024/k channels on k streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from tests import util

_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
C_, L, B = 1024, 1 << 20, 1 << 16
x = torch.empty((C_, L), dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
gbank.synth_fill(x, 1)
for k in (1, 2, 4):
    per = C_ // k
    banks = [gbank.FilterBank([gbank.iir_design(b, a)], [0] * per) for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    def run(nblocks):
        for j in range(nblocks):
            jj = j % (L // B)
            for i in range(k):
                off = (i * per * L + jj * B) * 8
                banks[i].updaten_ptr(y.data_ptr() + off, L, x.data_ptr() + off, L, B, streams[i])
    run(16); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(64); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d stream(s) x %4d channels: %.3f ms per 1024-channel block  %.0f GB/s" % (k, per, dt / 64 * 1e3, C_ * B * 16 * 64 / dt / 1e9))
    for bk in banks: bk.close()
