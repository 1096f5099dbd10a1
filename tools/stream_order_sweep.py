#!/usr/bin/env python3
"""RTFilterStream_updaten on one long stream, by design: time per pass and the split the library chose.
usage: stream_order_sweep.py [log2 n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank, stream as gstream
from tests import util
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << lg
x = torch.empty(n, dtype=torch.float64, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
s = torch.cuda.current_stream()
for key in ("bw8_bp_0.25_0.75", "bw4_bp_0.3_0.6", "bw5_lp_0.25"):
    _, b, a = util.design_coefs(key)
    fs = gstream.FilterStream(gbank.iir_design(b, a), device=0)
    for _ in range(2): fs.updaten_ptr(y.data_ptr(), x.data_ptr(), n, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5): fs.updaten_ptr(y.data_ptr(), x.data_ptr(), n, s)
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-18s 2^%d: %.3f ms  %.0f Msamples/s  %.0f GB/s  split %d  decay %d" % (key, lg, ms, n / ms / 1e3, n * 16 / ms / 1e6, fs.last_split, fs.decay_length), flush=True)
    fs.close()
