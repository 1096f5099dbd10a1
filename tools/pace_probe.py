#!/usr/bin/env python3
"""Paced tiled launches (RTFilterBank_set_pace): throughput without a schedule, with the library's period and with
fixed periods around it, by shape.  usage: pace_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from tests import util
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
x = torch.empty(1 << 30, dtype=torch.float64, device=dev).uniform_(-1, 1)
y = torch.empty_like(x)

def run(key, Cn, n, pace, nsteps):
    _, b, a = util.design_coefs(key)
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * Cn, device=0)
    bank.set_pace(pace)
    nb = max(1, min(16, x.numel() // (Cn * n)))
    def call(k):
        off = (k % nb) * Cn * n * 8
        bank.updaten_tiled_ptr(y.data_ptr() + off, x.data_ptr() + off, n, stream)
    for k in range(nsteps // 2): call(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for k in range(nsteps): call(k)
    e1.record(stream)
    torch.cuda.synchronize()
    bank.close()
    return Cn * n * nsteps / (e0.elapsed_time(e1) / 1e3) / 1e6

shapes = [("bw4_bp_0.3_0.6", 1024, 1 << 16), ("bw4_bp_0.3_0.6", 1024, 1 << 18), ("bw4_bp_0.3_0.6", 1024, 1 << 20),
          ("bw4_bp_0.3_0.6", 1024, 1 << 14), ("bw4_bp_0.3_0.6", 1024, 1 << 12), ("bw4_bp_0.3_0.6", 768, 1 << 16),
          ("bw4_bp_0.3_0.6", 512, 1 << 16), ("bw4_lp_0.5", 1024, 1 << 16), ("bw6_hp_0.3", 1024, 1 << 16),
          ("bw1_lp_0.1", 1024, 1 << 16)]
for key, Cn, n in shapes:
    nsteps = max(16, (1 << 22) // n)
    med = lambda p: float(np.median([run(key, Cn, n, p, nsteps) for _ in range(3)]))
    off, lib = med(0), med(-1)
    base = Cn * 32768.0 / 5950.0
    fixed = ["%.2f:%.0f" % (f, med(base * f)) for f in (0.92, 0.95, 0.98, 1.0, 1.03)]
    print("%-15s %4d ch x %7d: no schedule %.0f | library's period %.0f (%+.1f %%) | fixed, x %.0f ns: %s"
          % (key, Cn, n, off, lib, (lib / off - 1) * 100, base, "  ".join(fixed)), flush=True)
