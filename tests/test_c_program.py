"""A C caller of the reference API, compiled twice -- against this repository's include/RTFilter.h and (where the
reference tree exists) against the REFERENCE's own header -- and linked with this libdsp.so: KAT 1, 5 and 7 of
SURVEY.md 8(a) through the heap constructors, the variadic bank constructor, per-sample and block updates."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "dsp_amd", "lib")

PROGRAM = r"""
#include <stdio.h>
#include <RTFilter.h>

int main(void)
{
    double x[6] = {1, 2, 3, 4, 5, 6}, y[6];
    double b[3] = {0.2, 0.3, 0.1}, a[3] = {1.0, -0.5, 0.25};
    RTIIRFilter *f = RTIIRFilter_new(a, 3, b, 3, 0, NULL);
    if (!f) return 2;
    if (RTFilter_updaten(y, (RTFilter *)f, x, 6)) return 3;
    for (int i = 0; i < 6; ++i) printf("%a ", y[i]);
    printf("| %a %a %d\n", f->state[0], f->state[1], f->rtf.initialized);
    RTFilter_del((RTFilter *)f);

    double bf[3] = {0.5, 0.3, 0.2};
    RTFIRFilter *g = RTFIRFilter_new(bf, 3, 0, NULL);
    if (!g) return 4;
    for (int i = 0; i < 6; ++i) printf("%a ", RTFilter_update((RTFilter *)g, x[i]));
    printf("| %a %a\n", g->state[0], g->state[1]);
    RTFilter_del((RTFilter *)g);

    RTFIRFilter *m = RTFIRFilter_new_empty(4, 0, NULL);
    if (!m || moving_average(m, 3, NULL)) return 5;
    for (int i = 0; i < 6; ++i) printf("%a ", RTFilter_update((RTFilter *)m, x[i]));
    RTFilter_reset((RTFilter *)m);
    printf("| ");
    for (int i = 0; i < 6; ++i) printf("%a ", RTFilter_update((RTFilter *)m, x[i]));
    printf("\n");
    RTFilter_del((RTFilter *)m);
    return 0;
}
"""

EXPECTED = [
    "0x1p+0 0x1.0cccccccccccdp+0 0x1.9333333333334p+0 0x1.3666666666666p+1 0x1.a8cccccccccccp+1 0x1.09ccccccccccdp+2",
    [1.0, 1.5, 2.3, 3.3, 4.3, 5.3],
    ([1.0, 1.5, 2.0, 3.0, 3.9999999999999996, 5.0], [1.0, 4.5, 2.0, 3.0, 3.9999999999999996, 5.0]),
]


def _build_and_run(tmp_path, include_dir):
    src = tmp_path / "caller.c"
    exe = tmp_path / "caller"
    src.write_text(PROGRAM)
    cmd = ["gcc", "-std=c99", "-O2", "-I", include_dir, str(src), "-o", str(exe), "-L", LIBDIR, "-ldsp", "-lm",
           "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([LIBDIR, "/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    return r.stdout.strip().splitlines()


def _check(lines):
    assert len(lines) == 3
    iir, rest = lines[0].split("|")
    assert iir.strip() == EXPECTED[0]
    s0, s1, ini = rest.split()
    assert (float.fromhex(s0), float.fromhex(s1), int(ini)) == (3.546875, -0.43828124999999996, 0)
    fir, st = lines[1].split("|")
    assert [float.fromhex(v) for v in fir.split()] == EXPECTED[1]
    assert [float.fromhex(v) for v in st.split()] == [2.8, 1.2000000000000002]
    first, again = lines[2].split("|")
    assert [float.fromhex(v) for v in first.split()] == EXPECTED[2][0]
    assert [float.fromhex(v) for v in again.split()] == EXPECTED[2][1]        # stale state after RTFilter_reset


def test_c_caller_against_this_header(tmp_path):
    _check(_build_and_run(tmp_path, os.path.join(ROOT, "include")))


@pytest.mark.skipif(not os.path.isdir("/root/reference/include"), reason="reference tree only in the build container")
def test_c_caller_against_the_reference_header(tmp_path):
    _check(_build_and_run(tmp_path, "/root/reference/include"))


ACCESSOR_PROGRAM = r"""
#include <stdio.h>
#include <RTFilter.h>
#include <RTFilter_.h>
#include <dsputils.h>

int main(void)
{
    double b[3] = {0.2, 0.3, 0.1}, a[2] = {1.0, -0.5};
    RTIIRFilter *f = RTIIRFilter_new(a, 2, b, 3, 0, NULL);
    if (!f) return 2;
    printf("%zu %zu %a %a %d\n", RTIIRFilter_get_b_size(f), RTIIRFilter_get_a_size(f), RTIIRFilter_get_b(f)[2],
           RTIIRFilter_get_a(f)[1], (int)(RTIIRFilter_get_a(f) - RTIIRFilter_get_b(f)));
    printf("%zu %a %zu\n", IIRFilterBank_get_a_size(&f->ifb), IIRFilterBank_get_b(&f->ifb)[0], FIR_order(&f->ifb.fb));
    RTFilter_del((RTFilter *)f);
    double bf[2] = {0.5, 0.5};
    RTFIRFilter *g = RTFIRFilter_new(bf, 2, 0, NULL);
    if (!g) return 3;
    printf("%zu %a %zu\n", RTFIRFilter_get_b_size(g), RTFIRFilter_get_b(g)[1], FilterBank_get_b_size(&g->fb));
    RTFilter_del((RTFilter *)g);
    printf("%a %a %a %a\n", passband_ripple_epsilon_to_db(0.5), passband_ripple_db_to_epsilon(1.0),
           stopband_ripple_epsilon_to_db(0.1), stopband_ripple_db_to_epsilon(40.0));
    return 0;
}
"""


def _run_accessors(tmp_path, include_dir):
    global PROGRAM
    keep = PROGRAM
    try:
        PROGRAM = ACCESSOR_PROGRAM
        return _build_and_run(tmp_path, include_dir)
    finally:
        PROGRAM = keep


def test_accessor_header_of_this_library(tmp_path):
    """include/RTFilter_.h and include/dsputils.h (ref: include/RTFilter_.h:8-71, include/dsputils.h:1-12)."""
    lines = _run_accessors(tmp_path, os.path.join(ROOT, "include"))
    def parse(line):
        return [float.fromhex(t) if "x" in t else int(t) for t in line.split()]
    assert parse(lines[0]) == [3, 2, 0.1, -0.5, 3]
    assert parse(lines[1]) == [2, 0.2, 3]
    assert parse(lines[2]) == [2, 0.5, 2]
    vals = [float.fromhex(v) for v in lines[3].split()]
    assert abs(vals[0] - 0.9691001300805642) < 1e-15 and abs(vals[1] - 0.5088471399095875) < 1e-15
    assert abs(vals[2] - 20.043213737826427) < 1e-13 and abs(vals[3] - 0.010000500037503126) < 1e-16


@pytest.mark.skipif(not os.path.isdir("/root/reference/include"), reason="reference tree only in the build container")
def test_accessor_header_matches_the_reference_header(tmp_path):
    mine = _run_accessors(tmp_path, os.path.join(ROOT, "include"))
    (tmp_path / "ref").mkdir()
    theirs = _run_accessors(tmp_path / "ref", "/root/reference/include")
    assert mine == theirs


DEVICE_API_PROGRAM = r"""
/* a C99 caller of the additive device headers: compiles, links, and -- without a GPU -- gets clean refusals */
#include <stdio.h>
#include <RTFilterBank.h>
#include <RTFilterStream.h>
#include <RTFilterDesign.h>

static int my_allgather(void *ctx, const double *send_dev, double *recv_dev, size_t count, void *hip_stream)
{
    (void)ctx; (void)send_dev; (void)recv_dev; (void)count; (void)hip_stream;
    return 0;
}

int main(void)
{
    RTFilterStream_allgather_fn fn = my_allgather;
    size_t lens[2] = {4096, 4096};
    /* NULL stream handle: RTFB_EINVAL, nothing touches a device */
    int rc1 = RTFilterStream_split_updaten(NULL, NULL, NULL, lens, 0, 2, fn, NULL, NULL);
    int rc2 = RTFilterStream_split_end_state(NULL, 0, 2, fn, NULL, NULL);
    int rc3 = RTFilterStream_split_updaten_rccl(NULL, NULL, NULL, lens, 0, 2, NULL, NULL);
    int rc4 = RTFilterBank_updaten_tiled(NULL, NULL, NULL, RTFB_TILE, NULL);
    printf("%d %d %d %d %d\n", rc1 != 0, rc2 != 0, rc3 != 0, rc4 != 0, RTFB_TILE);
    return 0;
}
"""


def test_c_caller_of_the_device_headers(tmp_path):
    """include/RTFilterBank.h, RTFilterStream.h and RTFilterDesign.h are plain C: a C99 program that uses the
    round-3 entry points (tiled hand-over, the time split with a caller's all-gather) compiles against them, links
    with libdsp.so and is refused cleanly on NULL handles."""
    src = tmp_path / "device_caller.c"
    exe = tmp_path / "device_caller"
    src.write_text(DEVICE_API_PROGRAM)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", LIBDIR, "-ldsp", "-lm", "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + LIBDIR]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["1", "1", "1", "1", "2048"]
