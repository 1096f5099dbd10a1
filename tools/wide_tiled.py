#!/usr/bin/env python3
"""A wide IIR bank (config 4's IIR wave: 65536 channels x 64K, order 8) on dense rows and in the tiled block layout.
usage: wide_tiled.py [channels] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from tests import util
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
pitch = n + 128
x = torch.empty(Cn * pitch, dtype=torch.float64, device=dev).uniform_(-1, 1)
y = torch.empty_like(x)
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")

def timed(call, nsteps=4):
    call(); call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(nsteps): call()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nsteps

for rep in range(2):
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * Cn, device=0)
    ms = timed(lambda: bank.updaten_ptr(y.data_ptr(), pitch, x.data_ptr(), pitch, n, stream))
    print("dense rows, pitch n+128: %.3f ms  %.0f Msamples/s  %.2f TB/s" % (ms, Cn * n / ms / 1e3, Cn * n * 16 / ms / 1e9), flush=True)
    bank.close()
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * Cn, device=0)
    ms = timed(lambda: bank.updaten_tiled_ptr(y.data_ptr(), x.data_ptr(), n, stream))
    print("tiled [seg][ch][2048]:   %.3f ms  %.0f Msamples/s  %.2f TB/s" % (ms, Cn * n / ms / 1e3, Cn * n * 16 / ms / 1e9), flush=True)
    bank.close()
