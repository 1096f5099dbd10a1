#!/usr/bin/env python3
"""Where the two waves of the interleaved-segment kernel spend a step (build with -DDSP_P16_STAMPS): mean over the
workgroups of the time between milestones, per step, in microseconds."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsp_amd import bank as gbank
from dsp_amd import stream as gstream
from dsp_amd.cdsp import dspdll
from tests import util
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
dspdll.RTFilterStream_debug_exchange.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_ulonglong)]
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << lg
x = torch.empty((1, n), dtype=torch.float64, device="cuda")
gbank.synth_fill(x, 7)
x = x.view(-1)
y = torch.empty_like(x)
s = torch.cuda.current_stream()
for grid in (1024, 512, 256):
    fs = gstream.FilterStream(gbank.iir_design(b, a))
    fs.set_interleave(grid)
    for _ in range(3):
        fs(x, y, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    fs(x, y, s)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    G = fs.last_split
    nseg = n // 2048
    ptr, segs, gen = C.c_void_p(), C.c_size_t(), C.c_ulonglong()
    dspdll.RTFilterStream_debug_exchange(fs._h, C.byref(ptr), C.byref(segs), C.byref(gen))
    host = np.empty(G * 2 * 16, dtype=np.uint64)
    rc = hip.hipMemcpy(C.c_void_p(host.ctypes.data), C.c_void_p(ptr.value + nseg * 256), C.c_size_t(host.nbytes), C.c_int(2))
    assert rc == 0
    t = host.reshape(G, 2, 16).astype(np.float64) / 100.0 / (nseg / G)       # us per step
    print("grid %d: %.3f ms, %.2f us per step" % (G, ms, ms * 1e3 / (nseg / G)))
    names_f = ["publish+scan0", "wait back (b0)", "park (loads)", "wait (b1)", "pass 1", "up-sweep+hand-over"]
    names_b = ["stores", "wait front (b0)", "hbuf read + b1", "start state", "scan+down+transpose", "recurrence"]
    for role, names in ((0, names_f), (1, names_b)):
        m = t[:, role, :6].mean(axis=0)
        print("  %s: %s   sum %.2f" % ("front" if role == 0 else "back ", "  ".join("%s %.2f" % (nm, v) for nm, v in zip(names, m)), m.sum()))
    hw = host.reshape(G, 2, 16)[:, :, 10].astype(np.int64)
    xcc = host.reshape(G, 2, 16)[:, :, 11].astype(np.int64) & 15
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    se = (hw >> 13) & 7
    sh = (hw >> 12) & 1
    key = ((xcc[:, 0] * 8 + se[:, 0]) * 2 + sh[:, 0]) * 16 + cu[:, 0]
    print("  SIMD of (front, back) for the first workgroups:", [(int(simd[i, 0]), int(simd[i, 1])) for i in range(0, min(G, 1024), 97)])
    # per CU: which roles sit on which SIMD
    from collections import Counter, defaultdict
    percu = defaultdict(list)
    for i in range(G):
        percu[int(key[i])].append((i, int(simd[i, 0]), int(simd[i, 1])))
    pat = Counter()
    for k, lst in percu.items():
        f = Counter(x[1] for x in lst); bk = Counter(x[2] for x in lst)
        pat[(tuple(f.get(j, 0) for j in range(4)), tuple(bk.get(j, 0) for j in range(4)))] += 1
    print("  CUs seen %d; (fronts per SIMD, backs per SIMD) -> CUs: %s" % (len(percu), pat.most_common(6)))
    one = next(iter(percu.values()))
    print("  one CU: (workgroup, front SIMD, back SIMD):", one)
    fs.close()
