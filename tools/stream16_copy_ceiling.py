#!/usr/bin/env python3
"""What a plain copy reaches with config 5's access pattern: 2^30 / n rows of n samples a row apart (the sub-chunks of the
2^30 stream; 1024 workgroups resident at a time), one wave per row, 4096-sample pieces (iir_scan16_kernel's segments) -- against the same bytes swept in
the tiled order (1024 workgroups on one contiguous window per step).  usage: stream16_copy_ceiling.py [log2 sub-chunk length]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsp_amd.cdsp import dspdll
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nch, n = (1 << 30) >> lg, 1 << lg          # always 2^30 samples: rows = consecutive sub-chunks of one stream
x = torch.empty(nch * n, dtype=torch.float64, device="cuda").uniform_(-1, 1)
y = torch.empty_like(x)
s = torch.cuda.current_stream()
sp = C.c_void_p(s.cuda_stream)

def timed(call, reps=6):
    call(); call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): call()
    e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def show(what, ms):
    print("%-60s %.3f ms  %.0f GB/s" % (what, ms, nch * n * 16 / ms / 1e6), flush=True)

for threads, seg in ((64, 2048), (128, 4096), (128, 2048), (256, 4096), (256, 2048)):
    def call():
        rc = dspdll.dsp_copy_rows(C.c_void_p(y.data_ptr()), n, C.c_void_p(x.data_ptr()), n, nch, n, threads, seg, sp)
        assert rc == 0, rc
    show("rows a chunk apart, %d threads, pieces of %d" % (threads, seg), timed(call))
def tiles():
    rc = dspdll.dsp_copy_tiles(C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), nch, n, sp)
    assert rc == 0, rc
show("tiled order (contiguous window per step), 256 threads", timed(tiles))

# the same with half the footprint: in place
def inplace(threads, seg):
    def call():
        rc = dspdll.dsp_copy_rows(C.c_void_p(x.data_ptr()), n, C.c_void_p(x.data_ptr()), n, nch, n, threads, seg, sp)
        assert rc == 0, rc
    return call
show("rows a chunk apart, in place, 256 threads, pieces of 2048", timed(inplace(256, 2048)))
show("rows a chunk apart, in place, 64 threads, pieces of 2048", timed(inplace(64, 2048)))
# the filter itself (order 16, config 5's design) as a 1024-channel bank on the same rows, out of place and in place
from dsp_amd import bank as gbank
from tests import util
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * nch, device=0)
for _ in range(6): bank.updaten_ptr(y.data_ptr(), n, x.data_ptr(), n, n, s)          # (clocks settle under this load)
show("iir_scan16_kernel, out of place", timed(lambda: bank.updaten_ptr(y.data_ptr(), n, x.data_ptr(), n, n, s)))
x.uniform_(-1, 1)
bank.reset()
show("iir_scan16_kernel, in place", timed(lambda: bank.updaten_ptr(x.data_ptr(), n, x.data_ptr(), n, n, s), reps=3))
