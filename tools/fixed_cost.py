import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from dsp_amd.cdsp import dspdll
from dsp_amd import bank as gbank
from tests import util
fn = dspdll.dsp_copy_rows
L = 1 << 20; pitch = L + 128
xb = torch.empty((1024, pitch), dtype=torch.float64, device="cuda"); yb = torch.empty_like(xb)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(f, reps=10):
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[3:]))
for n in (2048, 8192, 65536):
    t = timeit(lambda: fn(C.c_void_p(yb.data_ptr()), pitch, C.c_void_p(xb.data_ptr()), pitch, 1024, n, 64, 2048, st))
    print("copy_rows 1024 x %6d (64 thr, seg 2048): %.4f ms" % (n, t))
t = timeit(lambda: None)
print("empty event pair: %.4f ms" % t)
_, b, a = util.design_coefs("bw4_bp_0.3_0.6")
for mode, nm in ((gbank.MODE_SCAN, "pipe"),):
    for var in ("0", "1", "2"):
        os.environ["DSP_AMD_SCAN_VARIANT"] = var
        bank = gbank.FilterBank([gbank.iir_design(b, a)], [0] * 1024, mode=mode)
        for n in (2048, 4096, 8192):
            t = timeit(lambda: bank.updaten_ptr(yb.data_ptr(), pitch, xb.data_ptr(), pitch, n))
            print("variant %s 1024 x %5d: %.4f ms" % (var, n, t))
        bank.close()
