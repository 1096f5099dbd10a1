#!/usr/bin/env python3
"""How to cut config 5's 2^30-sample stream into sub-chunks: C chunks of 2^30 / C samples as a C-channel order-16 bank
on the stream itself (row pitch = chunk length).  1024 long chunks keep 1024 streams 8 MiB apart in flight (one 2 MiB
page each, reads and writes); more, shorter chunks -- dispatched in order, 1024 resident at a time -- sweep a
contiguous window instead.  MODE=scan forces the scan kernel at every width (AUTO switches wide banks to the
lane-per-channel kernel).  usage: stream16_split_shape.py [log2 total]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank
from tests import util
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
s = torch.cuda.current_stream()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
total = 1 << lg
x = torch.empty(total, dtype=torch.float64, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
for lc in (20, 18, 17, 16, 15, 14):
    C, L = total >> lc, 1 << lc
    bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C, mode=gbank.MODE_SCAN)
    for _ in range(2):
        bank.updaten_ptr(y.data_ptr(), L, x.data_ptr(), L, L, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5):
        bank.updaten_ptr(y.data_ptr(), L, x.data_ptr(), L, L, s)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%6d chunks of 2^%d: %.3f ms  %.0f Msamples/s  %.0f GB/s" % (C, lc, ms, total / ms / 1e3, total * 16 / ms / 1e6), flush=True)
    bank.close()
