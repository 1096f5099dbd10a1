#!/usr/bin/env python3
"""Diagnostic: set-up time of banks with many DISTINCT designs: the batched host design
(dsp_design_batch) and RTFilterBank_from_designs (per-design scan tables + conditioning probe on host threads,
upload).  usage: bank_setup_time.py [n ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from dsp_amd import dsp

torch.zeros(1, device="cuda")
for n in [int(v) for v in sys.argv[1:]] or [1024, 4096, 65536, 262144]:
    rng = np.random.default_rng(3)
    wl = 0.25 + 0.1 * rng.random(n)
    wu = wl + 0.25 + 0.05 * rng.random(n)
    t0 = time.perf_counter()
    b, a, nc = dsp.design_batch("butterworth", 4, wl, wu)
    t1 = time.perf_counter()
    t2 = time.perf_counter()
    bank = gbank.FilterBank.from_arrays(gbank.KIND_IIR, b[:, :9], a[:, :9], device=0)
    t3 = time.perf_counter()
    x = torch.zeros((n, 4096), dtype=torch.float64, device="cuda")
    gbank.synth_fill(x, 1)
    y = bank(x)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("n = %6d distinct order-8 designs: design_batch %.3f s | %.3f s | "
          "RTFilterBank_from_designs %.3f s | first 4096-sample block %.3f s" % (n, t1 - t0, t2 - t1, t3 - t2, t4 - t3),
          flush=True)
    bank.close()
    del x, y
