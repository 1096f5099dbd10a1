#!/usr/bin/env python3
"""One variant of tools/chain_bench.py for a rocprofv3 --kernel-trace run:  chain_trace.py <variant> [steps]
   analysis:  chain_trace.py --analyze <dir>   (start / end of every pipelined-kernel launch relative to the first)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "--analyze":
    import csv
    import glob
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    pipe = [r for r in rows if "iir_pipe_kernel" in r[2]]
    pipe = pipe[-40:]
    t0 = pipe[0][0]
    prev = None
    for s, e, name, q in pipe:
        print("queue %s  start %9.1f  end %9.1f  dur %7.1f  start-to-start %7.1f  overlap with previous %7.1f us"
              % (q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev[0]) / 1e3 if prev else 0.0,
                 (prev[1] - s) / 1e3 if prev else 0.0))
        prev = (s, e)
    others = {}
    for s, e, name, q in rows:
        if "iir_pipe_kernel" not in name:
            others.setdefault(name[:60], []).append((e - s) / 1e3)
    for k, v in others.items():
        print("%-60s x%d  avg %.1f us" % (k, len(v), sum(v) / len(v)))
    sys.exit(0)

import torch  # noqa: E402

from dsp_amd import bank as gbank  # noqa: E402
from tests import util  # noqa: E402

variant = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
C, B, NB = 1024, 1 << 16, 16
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
dev = torch.device("cuda", 0)
tiled, chained = "tiled" in variant, "chained" in variant
x = torch.empty((NB, C, B), dtype=torch.float64, device=dev)
for j in range(NB):
    gbank.synth_fill(x[j], 0x5EED0002, chan0=0, n0=j * B)
y = torch.empty_like(x)
torch.cuda.synchronize()
stream = torch.cuda.current_stream()
bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * C, device=0)
for k in range(steps):
    j = k % NB
    if chained:
        bank.updaten_chained_ptr(y[j].data_ptr(), B, x[j].data_ptr(), B, B, tiled=tiled, stream=stream)
    elif tiled:
        bank.updaten_tiled_ptr(y[j].data_ptr(), x[j].data_ptr(), B, stream)
    else:
        bank.updaten_ptr(y[j].data_ptr(), B, x[j].data_ptr(), B, B, stream)
bank.chain_join(stream)
torch.cuda.synchronize()
bank.close()
print("done", variant)
