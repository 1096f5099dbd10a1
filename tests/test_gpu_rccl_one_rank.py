"""The RCCL code paths of the N > 1 legs executed once with ONE rank (backend "nccl", world_size 1) on the GPU box, so
that an 8-GPU node is not the first place they run: the wave-sized output gather of config 4 (bench.py --workload
mixed --gather, on its own stream, overlapped with the next wave) and shard.gather_rows / stream.split_update
under an initialised RCCL process group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611",
               BENCH_FORCE_DIST="1")
    return env


def test_bench_mixed_gather_runs_under_rccl_with_one_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mixed", "--gather",
                        "--mixed-total", "4096", "--mixed-channels", "1024", "--steps", "6", "--warmup", "2"],
                       env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["gather"] == "rows" and line["n_gpus"] == 1
    g = line["gather"]
    assert g["gather_ms_avg"] > 0 and g["filter_plus_gather_ms_per_step"] >= g["filter_ms_avg"] * 0.5
    assert line["value"] > 0


RUNNER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from dsp_amd import bank as gbank, shard, stream as gstream
from oracle import oracle as orc
from tests import util
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.arange(12.0, dtype=torch.float64, device="cuda").reshape(4, 3)
full = shard.gather_rows(x, 4)
assert torch.equal(full, x)
full2 = torch.empty_like(x)
shard.gather_rows_into(full2, x)
assert torch.equal(full2, x)
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
fs = gstream.FilterStream(gbank.iir_design(b, a))
xs = util.synth(3, 0, 0, 1 << 18)
y = gstream.split_update(fs, torch.from_numpy(xs).cuda(), total=1 << 18)
gstream.sync_end_state(fs)
want = orc.iir(b, a).updaten(xs)
assert np.max(np.abs(y.cpu().numpy() - want)) <= 1e-12 * np.max(np.abs(want))
dist.barrier()
dist.destroy_process_group()
print("ok")
"""


def test_shard_and_stream_helpers_under_rccl_with_one_rank():
    r = subprocess.run([sys.executable, "-c", RUNNER, ROOT], env=_env(), cwd=ROOT, capture_output=True, text=True,
                       timeout=200)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])


C_ADAPTER = r"""
import ctypes as C, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from dsp_amd import bank as gbank, stream as gstream
from dsp_amd.cdsp import dspdll
from oracle import oracle as orc
from tests import util
torch.cuda.init()
# a communicator of ONE rank, made with RCCL's own C API (what a C caller of the library holds)
rccl = None
for name in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so"):
    try:
        rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
        break
    except OSError:
        pass
assert rccl is not None, "no RCCL library"
uid = (C.c_char * 128)()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
u = Uid()
C.memmove(C.byref(u), uid, 128)
assert rccl.ncclCommInitRank(C.byref(comm), 1, u, 0) == 0
s = torch.cuda.current_stream()
send = torch.arange(40.0, dtype=torch.float64, device="cuda")
recv = torch.zeros_like(send)
rc = dspdll.dsp_debug_rccl_allgather(comm, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), 40,
                                     C.c_void_p(s.cuda_stream))
torch.cuda.synchronize()
assert rc == 0 and torch.equal(send, recv), rc
# the C entry itself with that communicator (world 1: the whole stream on this rank)
_, b, a = util.design_coefs("bw8_bp_0.25_0.75")
fs = gstream.FilterStream(gbank.iir_design(b, a))
xs = util.synth(3, 0, 0, 1 << 18)
x = torch.from_numpy(xs).cuda()
y = torch.empty_like(x)
lens = (C.c_size_t * 1)(x.numel())
rc = dspdll.RTFilterStream_split_updaten_rccl(fs._h, C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), lens, 0, 1,
                                              comm, C.c_void_p(s.cuda_stream))
assert rc == 0
assert dspdll.RTFilterStream_split_end_state_rccl(fs._h, 0, 1, comm, C.c_void_p(s.cuda_stream)) == 0
torch.cuda.synchronize()
want = orc.iir(b, a).updaten(xs)
assert np.max(np.abs(y.cpu().numpy() - want)) <= 1e-12 * np.max(np.abs(want))
# a world of two without a communicator is refused before anything is enqueued
lens2 = (C.c_size_t * 2)(x.numel(), x.numel())
assert dspdll.RTFilterStream_split_updaten_rccl(fs._h, C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), lens2, 0, 2,
                                                None, None) != 0
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print("ok")
"""


def test_c_split_entry_calls_rccl_through_its_adapter():
    """RTFilterStream_split_updaten_rccl (include/RTFilterStream.h): the adapter's ncclAllGather -- looked up at run
    time, datatype and argument order as rccl.h has them -- on a one-rank communicator made with RCCL's C API, and the
    entry point itself with that communicator."""
    r = subprocess.run([sys.executable, "-c", C_ADAPTER, ROOT], env=dict(os.environ), cwd=ROOT, capture_output=True,
                       text=True, timeout=200)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])
