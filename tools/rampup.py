#!/usr/bin/env python3
"""Diagnostic: config-2 step time as a function of how long the GPU has been busy since an idle second
(clock ramp of the device): events every 8 steps over 512 consecutive steps, twice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from tests import util
CH, BLOCK, NB = 1024, 65536, 16
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * CH)
x = torch.empty((NB, CH, BLOCK), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
for j in range(NB):
    gbank.synth_fill(x[j], 1, n0=j * BLOCK)
torch.cuda.synchronize()
s = torch.cuda.current_stream()
for rnd in range(3):
    time.sleep(1.0)
    G = 8
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(512 // G + 1)]
    ev[0].record(s)
    for k in range(512):
        bank.updaten_ptr(y[k % NB].data_ptr(), BLOCK, x[k % NB].data_ptr(), BLOCK, BLOCK, s)
        if (k + 1) % G == 0:
            ev[(k + 1) // G].record(s)
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) / G * 1e3 for i in range(len(ev) - 1)]
    print("round %d: us per step, groups of %d steps:" % (rnd, G), " ".join("%.0f" % v for v in t), flush=True)
