#!/usr/bin/env python3
"""RTFilterStream_updaten on config 5's stream, out of place and in place.  usage: stream_inplace.py [log2 n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank, stream as gstream
from tests import util
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << lg
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
x = torch.empty(n, dtype=torch.float64, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
s = torch.cuda.current_stream()
for what, out in (("out of place", y), ("in place", x)):
    fs = gstream.FilterStream(gbank.iir_design(b, a), device=0)
    for _ in range(6): fs.updaten_ptr(out.data_ptr(), x.data_ptr(), n, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(6): fs.updaten_ptr(out.data_ptr(), x.data_ptr(), n, s)
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 6
    print("%-14s 2^%d: %.3f ms  %.0f Msamples/s  %.0f GB/s  split %d" % (what, lg, ms, n / ms / 1e3, n * 16 / ms / 1e6, fs.last_split), flush=True)
    fs.close()
    x.uniform_(-1, 1)
