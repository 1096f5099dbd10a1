#!/usr/bin/env python3
"""Device time of the pieces of one multi-GPU config-5 step on one rank (2^27 samples of the 2^30 stream):
RTFilterStream_tail_state, export / import of the state pack, the rank's RTFilterStream_updaten.
usage: split_step_pieces.py [log2 n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd import bank as gbank, stream as gstream
from tests import util
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n = 1 << lg
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
fs = gstream.FilterStream(gbank.iir_design(b, a), device=0)
x = torch.empty(n, dtype=torch.float64, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
s = torch.cuda.current_stream()
pack = fs.new_pack()

def timed(what, call, reps=20):
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): call()
    e1.record(s); torch.cuda.synchronize()
    print("%-40s %.1f us" % (what, e0.elapsed_time(e1) / reps * 1e3), flush=True)

timed("updaten, 2^%d samples" % lg, lambda: fs.updaten_ptr(y.data_ptr(), x.data_ptr(), n, s))
timed("tail_state", lambda: fs.tail_state(x, pack, s))
timed("export_state", lambda: fs.export_state(pack, s))
timed("import_state", lambda: fs.import_state(pack, s))
