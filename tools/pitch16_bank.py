#!/usr/bin/env python3
"""Order-16 bank, 1024 channels x n: row pitch n (a power of two, what the contiguous time split of config 5 gives)
against padded pitches, iir_scan16_kernel and (DSP_AMD_PIPE16=1) the pipelined kernel.  usage: pitch16_bank.py [log2n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from tests import util
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
s = torch.cuda.current_stream()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 1 << lg
CH = 1024
x = torch.empty(CH * (N + 4096) , dtype=torch.float64, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
for pad in (0, 128, 2048, 2048 + 128, 0):
    ld = N + pad
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * CH)
    for _ in range(3):
        bank.updaten_ptr(y.data_ptr(), ld, x.data_ptr(), ld, N, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(6):
        bank.updaten_ptr(y.data_ptr(), ld, x.data_ptr(), ld, N, s)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 6
    print("pipe16=%s  1024 x 2^%d, pitch n + %4d: %.3f ms  %.0f Msamples/s  %.0f GB/s"
          % (os.environ.get("DSP_AMD_PIPE16", "0"), lg, pad, ms, CH * N / ms / 1e3, CH * N * 16 / ms / 1e6), flush=True)
    bank.close()
