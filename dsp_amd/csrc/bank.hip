/*
 * bank.hip -- host side of the MI355X path: the RTFilterBank object (the
 * block-streaming state manager), the device memory pool, and the C-ABI entry
 * points of include/RTFilterBank.h.
 *
 * A bank owns, in HBM, everything RTFilter_updaten touches per channel
 * (ref: include/RTFilter.h:30-59): the coefficient tables (deduplicated into
 * "designs"), the state vectors, filtered_value and initialized.  Samples are
 * never copied by this layer: callers hand device pointers, kernels read each
 * input once and write each output once.
 *
 * There is no CPU execution path in this file: if HIP or a gfx950 device is
 * missing every entry point fails with RTFB_EDEVICE.
 */
#include <dlfcn.h>
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "RTFilterBank.h"
#include "RTFilterDesign.h"
#include "RTFilterStream.h"
#include "dsp_internal.h"

/* Staging buffer of RTFilterBank_updaten_noalloc (the entry point for callers that cannot allocate: HIP graph
 * capture, serving loops, harnesses that forbid hipMalloc).  One copy per GPU the library runs on, held until the
 * process ends.  In host code this name is a stand-in, not a device address: host code uses the pointer that
 * hipGetSymbolAddress returned for the device (static_scratch_of), never the name. */
constexpr size_t kStaticScratchDoubles = (size_t)4 << 20;          /* 32 MiB */
static __device__ __align__(256) double scratch[kStaticScratchDoubles];

namespace {

using namespace dsp;

thread_local std::string t_last_error;

int fail(int code, const std::string &msg)
{
    t_last_error = msg;
    return code;
}

int fail_hip(hipError_t e, const char *what)
{
    return fail(RTFB_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                         \
    do {                                                      \
        hipError_t e_ = (expr);                               \
        if (e_ != hipSuccess) return fail_hip(e_, #expr);     \
    } while (0)

/* Make `device` current for the scope, restore the caller's device after
 * (the caller may be a framework that tracks its own current device). */
struct DeviceScope {
    int prev = -1;
    bool ok = false;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int check_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(RTFB_EDEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(RTFB_EINVAL, "device index out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RTFB_EDEVICE, std::string("device is ") + prop.gcnArchName +
                                      "; this library carries gfx950 (MI355X) code only");
    return RTFB_OK;
}

/* Device address of `scratch` on `device`, looked up once per device with that device current (every GPU holds
 * its own copy of the array at its own address).  Called at bank creation, never from the update path. */
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
double *g_scratch_ptr[kMaxDevices] = {};

int static_scratch_of(int device, double **ptr)
{
    if (device < 0 || device >= kMaxDevices) return fail(RTFB_EINVAL, "device index out of range");
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        DeviceScope ds(device);
        if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            (void)hipGetLastError();
            return fail_hip(e, "hipGetSymbolAddress(scratch)");
        }
        g_scratch_ptr[device] = static_cast<double *>(p);
    }
    *ptr = g_scratch_ptr[device];
    return RTFB_OK;
}

/* ------------------------------------------------------------------------ */
/* device memory pool: size-bucketed cache over hipMalloc                     */

struct Pool {
    std::mutex mu;
    struct Block { size_t bytes; int device; };
    /* a cached block; `ready` != nullptr: freed with dsp_pool_free_async, reusable once that event (recorded on the
     * freeing stream at the time of the free) has completed */
    struct Cached { void *ptr; hipEvent_t ready; };
    std::unordered_map<void *, Block> live;                  /* handed out */
    std::map<std::pair<int, size_t>, std::vector<Cached>> cache;   /* (device, bytes) -> free */
    std::unordered_map<int, size_t> in_use, cached;

    static size_t bucket(size_t bytes)
    {
        if (bytes == 0) bytes = 1;
        if (bytes <= (1u << 20)) {               /* up to 1 MiB: next power of two, >= 256 B */
            size_t b = 256;
            while (b < bytes) b <<= 1;
            return b;
        }
        const size_t g = size_t(2) << 20;        /* above: 2 MiB granules */
        return (bytes + g - 1) / g * g;
    }
};

Pool &pool()
{
    static Pool p;
    return p;
}

/* DSP_AMD_POOL_POISON=1 (diagnostic): every block is handed out filled with 0xFF bytes (NaN as doubles, -1 as
 * ints), so that a read of memory the library never wrote shows up in the results on every box, not only on one
 * whose memory happens to hold something else than zeros. */
static bool pool_poison()
{
    static const bool on = [] {
        const char *e = getenv("DSP_AMD_POOL_POISON");
        return e && e[0] == '1';
    }();
    return on;
}

static int pool_alloc_raw(void **out, size_t bytes, int device);

int pool_alloc(void **out, size_t bytes, int device)
{
    int rc = pool_alloc_raw(out, bytes, device);
    if (!rc && pool_poison()) {
        DeviceScope ds(device);
        (void)hipDeviceSynchronize();
        if (hipMemset(*out, 0xFF, Pool::bucket(bytes)) != hipSuccess) (void)hipGetLastError();
        (void)hipDeviceSynchronize();
    }
    return rc;
}

static int pool_alloc_raw(void **out, size_t bytes, int device)
{
    if (!out) return fail(RTFB_EINVAL, "dsp_pool_alloc: NULL result pointer");
    Pool &p = pool();
    const size_t b = Pool::bucket(bytes);
    {
        std::lock_guard<std::mutex> lk(p.mu);
        auto it = p.cache.find({device, b});
        if (it != p.cache.end()) {
            /* newest first; a block whose free is still pending on its stream is passed over */
            std::vector<Pool::Cached> &v = it->second;
            for (size_t k = v.size(); k-- > 0;) {
                if (v[k].ready) {
                    if (hipEventQuery(v[k].ready) != hipSuccess) {
                        (void)hipGetLastError();
                        continue;
                    }
                    (void)hipEventDestroy(v[k].ready);
                }
                void *ptr = v[k].ptr;
                v.erase(v.begin() + (long)k);
                p.cached[device] -= b;
                p.in_use[device] += b;
                p.live[ptr] = {b, device};
                *out = ptr;
                return RTFB_OK;
            }
        }
    }
    DeviceScope ds(device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    void *ptr = nullptr;
    hipError_t e = hipMalloc(&ptr, b);
    if (e == hipErrorOutOfMemory) {              /* give cached blocks back and retry once */
        (void)hipGetLastError();
        dsp_pool_trim(device);
        e = hipMalloc(&ptr, b);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? RTFB_ENOMEM : RTFB_EDEVICE,
                    std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(p.mu);
    p.live[ptr] = {b, device};
    p.in_use[device] += b;
    *out = ptr;
    return RTFB_OK;
}

/* Return a block to the cache.  `ordered`: the block may still be in use by work already enqueued on `stream`;
 * an event recorded there now gates its reuse (pool_alloc passes it over until the event has completed).
 * Without `ordered` the caller vouches that the device is done with the block (the bank's own frees come after
 * a stream or device synchronisation). */
int pool_free(void *ptr, bool ordered, hipStream_t stream)
{
    if (!ptr) return RTFB_OK;
    Pool &p = pool();
    Pool::Block blk;
    {
        std::lock_guard<std::mutex> lk(p.mu);
        auto it = p.live.find(ptr);
        if (it == p.live.end()) return fail(RTFB_EINVAL, "dsp_pool_free: pointer not from this pool");
        blk = it->second;
    }
    hipEvent_t ready = nullptr;
    if (ordered) {
        DeviceScope ds(blk.device);
        if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
        HIP_TRY(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ready, stream);
        if (e != hipSuccess) {
            (void)hipEventDestroy(ready);
            return fail_hip(e, "hipEventRecord");
        }
    }
    std::lock_guard<std::mutex> lk(p.mu);
    p.live.erase(ptr);
    p.in_use[blk.device] -= blk.bytes;
    p.cached[blk.device] += blk.bytes;
    p.cache[{blk.device, blk.bytes}].push_back({ptr, ready});
    return RTFB_OK;
}

/* ------------------------------------------------------------------------ */
/* host streaming: pinned staging rings around the device path                */

/* Two pinned host buffers and two device buffers each way, three streams.  Chunk k: the CPU stages the caller's
 * rows into pinned in[k&1] (while the GPU still works on chunk k-1), H2D on s_in, the filter on s_run (in order:
 * the state carries), D2H into pinned out[k&1] on s_out, and the CPU copies chunk k-1's pinned output to the
 * caller's rows meanwhile.  Pageable memory never meets the DMA engines, which then run at link speed. */
struct HostPipe {
    int device = -1;
    size_t bytes = 0;                        /* capacity of every buffer */
    double *h_in[2] = {nullptr, nullptr}, *h_out[2] = {nullptr, nullptr};
    double *d_in[2] = {nullptr, nullptr}, *d_out[2] = {nullptr, nullptr};
    hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
    hipEvent_t e_h2d[2] = {nullptr, nullptr}, e_run[2] = {nullptr, nullptr}, e_d2h[2] = {nullptr, nullptr};
    bool ready() const { return bytes != 0; }
};

int pool_alloc(void **out, size_t bytes, int device);
int pool_free(void *ptr, bool ordered = false, hipStream_t stream = nullptr);

void host_pipe_release(HostPipe &hp)
{
    if (hp.device < 0) return;
    DeviceScope ds(hp.device);
    if (hp.s_in) (void)hipStreamSynchronize(hp.s_in);
    if (hp.s_run) (void)hipStreamSynchronize(hp.s_run);
    if (hp.s_out) (void)hipStreamSynchronize(hp.s_out);
    for (int k = 0; k < 2; ++k) {
        if (hp.h_in[k]) (void)hipHostFree(hp.h_in[k]);
        if (hp.h_out[k]) (void)hipHostFree(hp.h_out[k]);
        if (hp.d_in[k]) (void)pool_free(hp.d_in[k], false, nullptr);
        if (hp.d_out[k]) (void)pool_free(hp.d_out[k], false, nullptr);
        if (hp.e_h2d[k]) (void)hipEventDestroy(hp.e_h2d[k]);
        if (hp.e_run[k]) (void)hipEventDestroy(hp.e_run[k]);
        if (hp.e_d2h[k]) (void)hipEventDestroy(hp.e_d2h[k]);
    }
    if (hp.s_in) (void)hipStreamDestroy(hp.s_in);
    if (hp.s_run) (void)hipStreamDestroy(hp.s_run);
    if (hp.s_out) (void)hipStreamDestroy(hp.s_out);
    hp = HostPipe();
}

/* All or nothing: resources are built into a local pipe and committed only when every one of them exists
 * (a half-built pipe is never left behind for the next call to trip over). */
int host_pipe_prepare(HostPipe &hp, int device, size_t bytes)
{
    if (hp.ready() && hp.device == device && hp.bytes >= bytes) return RTFB_OK;
    host_pipe_release(hp);
    HostPipe np;
    np.device = device;
    int rc = RTFB_OK;
    auto hipok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) rc = fail_hip(e, what);
        return e == hipSuccess;
    };
    hipok(hipStreamCreateWithFlags(&np.s_in, hipStreamNonBlocking), "hipStreamCreate") &&
        hipok(hipStreamCreateWithFlags(&np.s_run, hipStreamNonBlocking), "hipStreamCreate") &&
        hipok(hipStreamCreateWithFlags(&np.s_out, hipStreamNonBlocking), "hipStreamCreate");
    for (int k = 0; k < 2 && !rc; ++k) {
        hipok(hipEventCreateWithFlags(&np.e_h2d[k], hipEventDisableTiming), "hipEventCreate") &&
            hipok(hipEventCreateWithFlags(&np.e_run[k], hipEventDisableTiming), "hipEventCreate") &&
            hipok(hipEventCreateWithFlags(&np.e_d2h[k], hipEventDisableTiming), "hipEventCreate") &&
            hipok(hipHostMalloc(reinterpret_cast<void **>(&np.h_in[k]), bytes, hipHostMallocDefault), "hipHostMalloc") &&
            hipok(hipHostMalloc(reinterpret_cast<void **>(&np.h_out[k]), bytes, hipHostMallocDefault), "hipHostMalloc");
        void *p = nullptr;
        if (!rc && !(rc = pool_alloc(&p, bytes, device))) np.d_in[k] = static_cast<double *>(p);
        if (!rc && !(rc = pool_alloc(&p, bytes, device))) np.d_out[k] = static_cast<double *>(p);
    }
    if (rc) {
        const std::string keep = t_last_error;
        host_pipe_release(np);
        t_last_error = keep;
        return rc;
    }
    np.bytes = bytes;
    hp = np;
    return RTFB_OK;
}

/* Strided row copy on the host, split over a few threads when it is large (one thread moves ~6 GB/s, the link
 * takes several times that). */
void copy_rows_host(double *dst, size_t ld_dst, const double *src, size_t ld_src, size_t rows, size_t len)
{
    const size_t bytes = rows * len * sizeof(double);
    size_t nthreads = std::min<size_t>({(size_t)8, (size_t)std::max(1u, std::thread::hardware_concurrency() / 2),
                                        bytes >> 22});
    if (nthreads <= 1) {
        for (size_t r = 0; r < rows; ++r) std::memcpy(dst + r * ld_dst, src + r * ld_src, len * sizeof(double));
        return;
    }
    /* split by rows when there are enough of them, else by columns of the one long row */
    std::vector<std::thread> th;
    for (size_t t = 0; t < nthreads; ++t) {
        th.emplace_back([=]() {
            if (rows >= nthreads) {
                for (size_t r = rows * t / nthreads; r < rows * (t + 1) / nthreads; ++r)
                    std::memcpy(dst + r * ld_dst, src + r * ld_src, len * sizeof(double));
            } else {
                const size_t c0 = len * t / nthreads, c1 = len * (t + 1) / nthreads;
                for (size_t r = 0; r < rows; ++r)
                    std::memcpy(dst + r * ld_dst + c0, src + r * ld_src + c0, (c1 - c0) * sizeof(double));
            }
        });
    }
    for (std::thread &x : th) x.join();
}

/* rows x n doubles between the caller's host arrays (row pitches ld_in / ld_out) and the device path, `nt`
 * samples of every row per chunk.  filter(d_out, d_in, len, stream) advances all rows by `len` samples on
 * dense device blocks of row pitch `len`.  Blocks until `out` is complete; on an error the streams are drained
 * before returning (no copy is left in flight towards the caller's memory). */
template <typename F>
int host_pipe_run(HostPipe &hp, size_t rows, size_t n, const double *in, size_t ld_in, double *out, size_t ld_out,
                  size_t nt, F &&filter)
{
    int rc = RTFB_OK;
    auto chk = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) rc = fail_hip(e, what);
        return rc == RTFB_OK;
    };
    auto drain = [&](size_t k, size_t pos, size_t len) {          /* chunk k: pinned out -> caller */
        const int slot = (int)(k & 1);
        if (!chk(hipEventSynchronize(hp.e_d2h[slot]), "hipEventSynchronize")) return;
        copy_rows_host(out + pos, ld_out, hp.h_out[slot], len, rows, len);
    };
    size_t k = 0, prev_pos = 0, prev_len = 0;
    for (size_t pos = 0; pos < n && !rc; pos += nt, ++k) {
        const size_t len = std::min(nt, n - pos);
        const int slot = (int)(k & 1);
        if (k >= 2 && !chk(hipEventSynchronize(hp.e_h2d[slot]), "hipEventSynchronize")) break;   /* pinned in[slot] is free */
        copy_rows_host(hp.h_in[slot], len, in + pos, ld_in, rows, len);
        if (k >= 2) chk(hipStreamWaitEvent(hp.s_in, hp.e_run[slot], 0), "hipStreamWaitEvent");
        chk(hipMemcpyAsync(hp.d_in[slot], hp.h_in[slot], rows * len * sizeof(double), hipMemcpyHostToDevice, hp.s_in),
            "hipMemcpyAsync");
        chk(hipEventRecord(hp.e_h2d[slot], hp.s_in), "hipEventRecord");
        chk(hipStreamWaitEvent(hp.s_run, hp.e_h2d[slot], 0), "hipStreamWaitEvent");
        if (k >= 2) chk(hipStreamWaitEvent(hp.s_run, hp.e_d2h[slot], 0), "hipStreamWaitEvent");
        if (!rc) rc = filter(hp.d_out[slot], hp.d_in[slot], len, hp.s_run);
        chk(hipEventRecord(hp.e_run[slot], hp.s_run), "hipEventRecord");
        chk(hipStreamWaitEvent(hp.s_out, hp.e_run[slot], 0), "hipStreamWaitEvent");
        chk(hipMemcpyAsync(hp.h_out[slot], hp.d_out[slot], rows * len * sizeof(double), hipMemcpyDeviceToHost, hp.s_out),
            "hipMemcpyAsync");
        chk(hipEventRecord(hp.e_d2h[slot], hp.s_out), "hipEventRecord");
        if (k >= 1 && !rc) drain(k - 1, prev_pos, prev_len);       /* overlaps the GPU's work on chunk k */
        prev_pos = pos;
        prev_len = len;
    }
    if (!rc && k >= 1) drain(k - 1, prev_pos, prev_len);
    if (rc) {
        const std::string keep = t_last_error;
        (void)hipStreamSynchronize(hp.s_in);
        (void)hipStreamSynchronize(hp.s_run);
        (void)hipStreamSynchronize(hp.s_out);
        t_last_error = keep;
    }
    return rc;
}

constexpr size_t kHostChunkBytes = (size_t)64 << 20;      /* per staging buffer */

/* typed helper: pooled device array filled from a host vector */
template <typename T>
int upload(T **dev, const std::vector<T> &host, int device)
{
    *dev = nullptr;
    if (host.empty()) return RTFB_OK;
    void *p = nullptr;
    int rc = pool_alloc(&p, host.size() * sizeof(T), device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<T *>(p);
    return RTFB_OK;
}

/* ------------------------------------------------------------------------ */
/* bank                                                                       */

struct Design {
    int kind;
    int nb, na;
    std::vector<double> b, a;
    int coef_off = 0;
    int pow_off = 0;
    int pow2_off = 0;
    int levels2 = 0;
    int gtab_off = 0;
    int ttab_off = -1;
    int levels1 = kScanLevels;
    int gtab1_off = 0;
    double scan_error = HUGE_VAL;       /* conditioning probe result (fast IIR designs) */
    bool has_tables = false;            /* scan tables built (false: the design's group is sequential-only) */
    std::vector<int> section_design;    /* RTFB_KIND_SOS, not fused: the IIR design (nb = na = 3) of every section */
    bool sos_fused = false;             /* RTFB_KIND_SOS with <= kMaxFusedSections sections, every a0 == 1: all sections
                                           per sample in ONE pass (2 states per section on the pipelined kernel, or the
                                           lane-per-channel kernel); other cascades run one pass per section */
    bool scan_safe() const;
    /* SOS: three slots per section, laid out section after section (the third, never written, is 0) */
    int state_slots() const { return kind == RTFB_KIND_IIR ? std::max(nb, na) : nb; }
    bool fast_iir() const
    {
        return kind == RTFB_KIND_IIR && nb == na && nb >= 2 && nb - 1 <= kMaxFastOrder;
    }
    bool fast() const { return fast_iir() || sos_fused; }
    /* states of the system the fast kernels run: the filter order, or two per fused section */
    int fast_n() const { return sos_fused ? 2 * (nb / 3) : nb - 1; }
};

enum GroupClass { kIirFast = 0, kIirGeneric = 1, kFir = 2 };

struct Group {
    GroupClass cls;
    int order = 0;                      /* kIirFast: N */
    bool scan_safe = false;             /* kIirFast: every design passed the conditioning probe */
    bool has_tables = true;             /* kIirFast: every design carries its scan tables          */
    bool unit_a0 = false;               /* kIirFast: every design has a[0] == 1.0               */
    bool even_b = true;                 /* kIirFast: every design has b[odd] == 0 exactly       */
    int sos_sections = 0;               /* kIirFast: > 0 = fused cascades of this many sections (order = 2 x that) */
    std::vector<int32_t> chans;         /* global channel ids, ascending */
    std::vector<int32_t> chan_design;   /* design index per entry of chans (a section's design for SOS channels) */
    int stage = 0;                      /* SOS: which section of its channels this group runs; stage k reads the
                                           output rows that stage k-1 wrote (in place), stage 0 the input rows */
    int state_ld = 1;
    /* device */
    int32_t *d_chan = nullptr;
    int32_t *d_design = nullptr;
    double *d_state = nullptr;
    double *d_state_alt = nullptr;      /* FIR: where the end-state kernels build the next state before it is
                                           committed to d_state (never holds state between calls) */
    int32_t *d_aux = nullptr;           /* FIR: prologue -> main/epilogue hand-off */
    int kmax = 0;                       /* FIR: longest filter in the group        */
    bool has_partial = false;           /* FIR: a partial_init design is present    */
    int single_design = -2;             /* design shared by all channels; -1 = several; -2 = none seen yet */
    int levels2_max = 0;                /* fast IIR: most combine levels any design needs */
    double *d_scratch = nullptr;        /* FIR, in-place calls: private copy of the input */
    size_t scratch_doubles = 0;
    double *d_fval = nullptr;
    int32_t *d_init = nullptr;
    bool contiguous = false;
    double pace_ns = -1.0;              /* paced tiled launches: period per segment set by RTFilterBank_set_pace
                                           (0 = off, < 0 = the library's choice, pace_period) */
};

/* RTFilterBank_updaten_chained: two internal streams, a ring of events and the device words of ChainView
 * (dsp_internal.h).  Built by the first chained call of a bank, released with the bank. */
constexpr int kChainRing = 4;
struct ChainState {
    bool ready = false;                 /* device words below exist                                           */
    bool have_events = false;           /* streams, events and the pinned error word exist                    */
    bool open = false;                  /* links are in flight: the group's plain state arrays are stale       */
    hipStream_t s[2] = {nullptr, nullptr};
    hipEvent_t done[kChainRing] = {}, in_ready[kChainRing] = {};
    uint64_t seq = 0;                   /* chained launches of this bank so far; generation of the next = seq + 1 */
    uint64_t links = 0;                 /* launches of the chain that is open                                  */
    int count = 0, order = 0;
    unsigned long long *d_queue = nullptr;      /* kChainRing x count ready-queue entries                     */
    unsigned long long *d_counters = nullptr;   /* finished[slot] at 16 * slot, started[slot] at 16 * (ring + slot): a
                                                   128-byte line each                                          */
    unsigned int *d_error = nullptr;
    unsigned int *h_error = nullptr;            /* pinned: the error word as of the last join                  */
    double *d_records = nullptr;
    unsigned long long finished_total[kChainRing] = {}, started_total[kChainRing] = {};
};

}  // namespace

struct RTFilterBank {
    ChainState chain;
    int device = 0;
    size_t nch = 0;
    int mode = RTFB_MODE_AUTO;
    std::vector<Design> designs;
    std::vector<int32_t> design_of;     /* per channel */
    std::vector<Group> groups;
    std::vector<RTFilter *> host_filters;   /* from_filters only */
    size_t state_ld = 1;                /* uniform ld of the get/set_state API */
    /* device tables shared by all groups */
    int32_t *d_coef_off = nullptr, *d_pow_off = nullptr, *d_pow2_off = nullptr, *d_levels2 = nullptr, *d_gtab_off = nullptr, *d_ttab_off = nullptr,
            *d_levels1 = nullptr, *d_gtab1_off = nullptr,
            *d_nb = nullptr, *d_na = nullptr, *d_kind = nullptr;
    double *d_coefs = nullptr, *d_pows = nullptr;
    double *d_static_scratch = nullptr;   /* this device's copy of `scratch` (RTFilterBank_updaten_noalloc) */
    int stream_rows = 0;                  /* set by RTFilterStream around one call: the rows are consecutive sub-chunks
                                             of one stream, the order 9..16 scan kernel runs its own run-ups
                                             (BlockIO::runup: 4 = over the whole segment in front of a row, 1 = over
                                             its last 1024 samples) */
    const double *stream_runup_src = nullptr;   /* with it (in-place streams): BlockIO::runup_src */
    HostPipe host;                        /* staging rings of RTFilterBank_updaten_host, built on first use */
};

namespace {

/* A^(T * 2^k), k = 0..5, for the homogeneous recurrence s' = A s of
 * RTIIRFilter_update with x = 0 (ref: src/RTFilter.c:417-433):
 *   y = s0 / a0 ;  s_i' = s_{i+1} - a_{i+1} * y   =>  A = U - (a_{1..N}/a0) e0^T.
 * Computed in long double, stored as doubles, row-major. */
void scan_powers(const Design &d, int T, int levels, std::vector<double> &out)
{
    const int N = d.nb - 1;
    typedef std::vector<long double> Mat;
    auto mul = [N](const Mat &p, const Mat &q) {
        Mat r((size_t)N * N, 0.0L);
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < N; ++k) {
                const long double pik = p[(size_t)i * N + k];
                if (pik == 0.0L) continue;
                for (int j = 0; j < N; ++j) r[(size_t)i * N + j] += pik * q[(size_t)k * N + j];
            }
        return r;
    };
    Mat A((size_t)N * N, 0.0L);
    for (int i = 0; i < N; ++i) {
        A[(size_t)i * N] = -(long double)d.a[i + 1] / (long double)d.a[0];
        if (i + 1 < N) A[(size_t)i * N + i + 1] += 1.0L;
    }
    Mat P = A;                                  /* A^1 -> A^T by squaring (T is a power of two) */
    for (int t = 1; t < T; t <<= 1) P = mul(P, P);
    for (int lvl = 0; lvl < levels; ++lvl) {
        for (size_t e = 0; e < P.size(); ++e) out.push_back((double)P[e]);
        P = mul(P, P);
    }
}

/* Zero-state response table of the two-wave scan kernel.  With s' = A s + B x
 * (B_i = b_{i+1} - a_{i+1} b0 / a0) the state a chunk of T samples leaves behind when it starts
 * from zero is  sum_n A^(T-1-n) B x[n]:  N multiply-adds per sample instead of the 2N+1 of
 * running the recurrence.  Gc is the same for a constant fed into the last delay slot (the
 * reference's never-written state[N]).  Long double, stored as doubles: G[n][i] then Gc[i]. */
void scan2_gtable(const Design &d, int T, std::vector<double> &out)
{
    const int N = d.nb - 1;
    std::vector<long double> alpha(N), w(N), acc(N, 0.0L), e(N, 0.0L);
    const long double a0 = d.a[0], b0 = d.b[0];
    for (int i = 0; i < N; ++i) {
        alpha[i] = (long double)d.a[i + 1] / a0;
        w[i] = (long double)d.b[i + 1] - alpha[i] * b0;
    }
    auto advance = [&](std::vector<long double> &z) {       /* z <- A z */
        const long double z0 = z[0];
        for (int i = 0; i < N; ++i) z[i] = (i + 1 < N ? z[i + 1] : 0.0L) - alpha[i] * z0;
    };
    std::vector<double> table((size_t)T * N);
    for (int n = T - 1; n >= 0; --n) {                       /* w = A^(T-1-n) B */
        for (int i = 0; i < N; ++i) table[(size_t)n * N + i] = (double)w[i];
        advance(w);
    }
    e[N - 1] = 1.0L;
    for (int m = 0; m < T; ++m) {
        for (int i = 0; i < N; ++i) acc[i] += e[i];
        advance(e);
    }
    out.insert(out.end(), table.begin(), table.end());
    for (int i = 0; i < N; ++i) out.push_back((double)acc[i]);
}

/* Row propagators of the pipelined kernel: A^(32 j), j = 0..63 (long double products), stored [e][j]. */
void pipe_ttable(const Design &d, int T, std::vector<double> &out)
{
    const int N = d.nb - 1;
    typedef std::vector<long double> Mat;
    auto mul = [N](const Mat &p, const Mat &q) {
        Mat r((size_t)N * N, 0.0L);
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < N; ++k)
                for (int j = 0; j < N; ++j) r[(size_t)i * N + j] += p[(size_t)i * N + k] * q[(size_t)k * N + j];
        return r;
    };
    Mat A((size_t)N * N, 0.0L), P((size_t)N * N, 0.0L);
    for (int i = 0; i < N; ++i) {
        A[(size_t)i * N] = -(long double)d.a[i + 1] / (long double)d.a[0];
        if (i + 1 < N) A[(size_t)i * N + i + 1] += 1.0L;
        P[(size_t)i * N + i] = 1.0L;
    }
    Mat AT = A;
    for (int t = 1; t < T; t <<= 1) AT = mul(AT, AT);
    /* element-major: entry e of all 64 propagators is contiguous, so that the lanes of the back wave
     * (lane j holds A^(32 j)) read 512 consecutive bytes per load instead of 64 separate lines */
    const size_t base = out.size();
    out.resize(base + (size_t)kWave * N * N);
    for (int j = 0; j < kWave; ++j) {
        for (size_t e = 0; e < P.size(); ++e) out[base + e * kWave + j] = (double)P[e];
        P = mul(AT, P);
    }
}

/* ---- fused second-order sections: the cascade as one linear system ---------------------------------------------
 * z = (s^1_0, s^1_1, s^2_0, s^2_1, ...), z' = A z + B x with A block lower-triangular: section k sees the output
 * of section k-1 within the same sample.  A and B are read off one step of the cascade in long double (columns of
 * A: the step from a unit state with x = 0; B: the step from the zero state with x = 1). */
struct LinSys {
    int N = 0;
    std::vector<long double> A, B;
};

LinSys sos_system(const Design &d)
{
    const int S = d.nb / 3, N = 2 * S;
    LinSys sys;
    sys.N = N;
    sys.A.assign((size_t)N * N, 0.0L);
    sys.B.assign(N, 0.0L);
    auto step = [&](std::vector<long double> z, long double x) {
        for (int k = 0; k < S; ++k) {
            const long double b0 = d.b[3 * k], b1 = d.b[3 * k + 1], b2 = d.b[3 * k + 2];
            const long double a0 = d.a[3 * k], a1 = d.a[3 * k + 1], a2 = d.a[3 * k + 2];
            const long double y = (b0 * x + z[2 * k]) / a0;
            z[2 * k] = z[2 * k + 1] + (b1 * x - a1 * y);
            z[2 * k + 1] = b2 * x - a2 * y;
            x = y;
        }
        return z;
    };
    for (int j = 0; j < N; ++j) {
        std::vector<long double> e(N, 0.0L);
        e[j] = 1.0L;
        const std::vector<long double> col = step(e, 0.0L);
        for (int i = 0; i < N; ++i) sys.A[(size_t)i * N + j] = col[i];
    }
    sys.B = step(std::vector<long double>(N, 0.0L), 1.0L);
    return sys;
}

std::vector<long double> sys_mul(int N, const std::vector<long double> &p, const std::vector<long double> &q)
{
    std::vector<long double> r((size_t)N * N, 0.0L);
    for (int i = 0; i < N; ++i)
        for (int k = 0; k < N; ++k) {
            const long double pik = p[(size_t)i * N + k];
            if (pik == 0.0L) continue;
            for (int j = 0; j < N; ++j) r[(size_t)i * N + j] += pik * q[(size_t)k * N + j];
        }
    return r;
}

/* A^(T 2^k), k < levels (as scan_powers) */
void sys_powers(const LinSys &sys, int T, int levels, std::vector<double> &out)
{
    std::vector<long double> P = sys.A;
    for (int t = 1; t < T; t <<= 1) P = sys_mul(sys.N, P, P);
    for (int lvl = 0; lvl < levels; ++lvl) {
        for (long double v : P) out.push_back((double)v);
        P = sys_mul(sys.N, P, P);
    }
}

/* G[n][i] = (A^(T-1-n) B)_i, n < T, then N zeros where a transfer-function design keeps the response to its
 * never-written tail slot (a cascade has none) -- the layout of scan2_gtable */
void sys_gtable(const LinSys &sys, int T, std::vector<double> &out)
{
    const int N = sys.N;
    std::vector<long double> w = sys.B, t(N);
    std::vector<double> table((size_t)T * N);
    for (int n = T - 1; n >= 0; --n) {
        for (int i = 0; i < N; ++i) table[(size_t)n * N + i] = (double)w[i];
        for (int i = 0; i < N; ++i) {
            long double acc = 0.0L;
            for (int j = 0; j < N; ++j) acc += sys.A[(size_t)i * N + j] * w[j];
            t[i] = acc;
        }
        w = t;
    }
    out.insert(out.end(), table.begin(), table.end());
    out.insert(out.end(), (size_t)N, 0.0);
}

/* row propagators A^(T j), j < 64, stored [e][j] (as pipe_ttable) */
void sys_ttable(const LinSys &sys, int T, std::vector<double> &out)
{
    const int N = sys.N;
    std::vector<long double> AT = sys.A, P((size_t)N * N, 0.0L);
    for (int t = 1; t < T; t <<= 1) AT = sys_mul(N, AT, AT);
    for (int i = 0; i < N; ++i) P[(size_t)i * N + i] = 1.0L;
    const size_t base = out.size();
    out.resize(base + (size_t)kWave * N * N);
    for (int j = 0; j < kWave; ++j) {
        for (size_t e = 0; e < P.size(); ++e) out[base + e * kWave + j] = (double)P[e];
        P = sys_mul(N, AT, P);
    }
}

/* Combine levels of the two-wave scan kernel that still matter for a design:
 * level k applies A^(32*2^k); once every entry of that power (and of all
 * higher ones) is below 2^-66 its contribution is more than 2^-13 below one ulp
 * of the state it is added to -- four orders of magnitude under the few ulp that
 * re-association costs these kernels anyway -- and the kernel skips it.  (2^-75
 * until round 3: for config 5's design max|A^512| = 2^-70.2 kept a fourth level
 * of the order 9..16 kernel alive, 16 of 128 matrix instructions per segment in
 * a kernel that is bounded by FP64 arithmetic: 303 000 -> 311 000 Msamples/s.) */
constexpr double kScan2Negligible = 0x1.0p-66;
constexpr size_t kMaxPipeDesigns = 1024;     /* 64 propagators of N x N doubles per design */

int scan2_levels(const double *pw, int N, int nlevels = kScan2Levels)
{
    int need = 0;
    for (int lvl = 0; lvl < nlevels; ++lvl) {
        double m = 0.0;
        for (int e = 0; e < N * N; ++e) {
            const double v = std::fabs(pw[(size_t)lvl * N * N + e]);
            if (!(v <= m)) m = v;                /* NaN counts as large */
        }
        if (!(m < kScan2Negligible)) need = lvl + 1;
    }
    return need;
}

/* Conditioning probe for the scan kernels (host, bank-creation time only; it
 * produces no output sample).  The scan kernels re-associate the recurrence
 * (zero-state pass from the response table G, prefix combine with A^(T 2^k),
 * carried state through row propagators or a virtual row), so their distance
 * from the reference's sequential arithmetic depends on the design's conditioning
 * (SURVEY.md 0.5): ~1e-16 for the well-conditioned designs, 1e-8 and worse for
 * narrow-band / high-Q ones.  This runs THE ALGORITHM OF THE KERNEL THAT WILL
 * EXECUTE -- same row length T, same rows per segment, same tables (the very
 * doubles the kernel reads), same order of the fused multiply-adds, same skipped
 * levels -- next to the plain recurrence on two segments of pseudo-random input
 * and returns max|dy| / max|y|.  (The order 9..16 kernel accumulates its two
 * table-driven stages on the matrix cores, four terms per instruction; here they
 * are one FMA chain in the same term order.)  RTFB_MODE_AUTO sends a design to
 * its scan kernel only if the probe stays below kScanSafeError; otherwise its
 * channels run on the bit-exact sequential kernel.
 * The input is zero-mean noise of unit scale: the bound that AUTO inherits is
 * relative to the larger of the output peak and the input scale seen through
 * the filter's state (a stream dominated by DC or out-of-band content that the
 * design rejects keeps the states at the input's scale while the output is
 * small; re-association error stays relative to the former). */
constexpr double kScanSafeError = 1e-13;     /* 10x margin under the 1e-12 parity bar */

/* orders 9..16: iir_scan16_kernel serves every shape.  The two-wave pipelined kernel (pipe16_kernel.hip) is an
 * alternative for launches of whole 2048-sample segments, OFF by default: measured, it is no faster (DESIGN.md 4b:
 * on this part the FP64 matrix instructions and the FP64 vector ALU draw on the same arithmetic rate, so splitting
 * the stages over two waves overlaps nothing that the rate does not already bound).  DSP_AMD_PIPE16=1 turns it on
 * for banks; RTFilterStream_set_interleave(stream, workgroups > 0) for a stream. */
constexpr size_t kPipe16Seg = 2048;
bool pipe16_enabled()
{
    static const bool on = [] {
        const char *e = getenv("DSP_AMD_PIPE16");
        return e && e[0] == '1';
    }();
    return on;
}

bool Design::scan_safe() const { return scan_error <= kScanSafeError; }

struct ProbeKernel {
    int T;                  /* samples per row                               */
    int R;                  /* rows per segment                              */
    bool propagators;       /* carried state by A^(T j) per row (pipelined kernel) instead of a virtual row -1 */
};
constexpr ProbeKernel kProbePipe = {kScan2T, kWave, true};         /* iir_pipe_kernel, orders <= 8, <= 1024 designs */
constexpr ProbeKernel kProbeScan2 = {kScan2T, kScan2Rows, false};  /* iir_scan2_kernel, orders <= 8, more designs   */
constexpr ProbeKernel kProbeScan16 = {kScanT, kWave, false};       /* iir_scan16_kernel, orders 9..16                */

/* pw: nlev_table matrices A^(T 2^k), row-major; gtab: G[n][i], n < T, then Gc[i]; ttab: [e][j] (R = 64) or null */
/* NN: the order as a compile-time constant (0 = any): the loops over the states then unroll and vectorise --
 * the probe is what a bank of many distinct designs spends its set-up time in. */
template <bool kHwFma, int NN>
static inline __attribute__((always_inline)) double scan_probe_body(const Design &d, const ProbeKernel &pk,
                                                                    const double *pw, int nlev, const double *gtab,
                                                                    const double *ttab)
{
    const int N = NN ? NN : d.fast_n(), T = pk.T, R = pk.R, L = 2 * R * T, S = N + 1;
    const bool sos = d.sos_fused;       /* cascade: reference = the sections one after the other per sample */
    std::vector<double> x(L), yref(L), y(L);
    uint64_t z = 0x243F6A8885A308D3ull;
    for (int k = 0; k < L; ++k) {                /* splitmix64 stream in [-1, 1) */
        z += 0x9E3779B97F4A7C15ull;
        uint64_t h = z;
        h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
        h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
        h ^= h >> 31;
        x[k] = (double)(int64_t)(h >> 11) * 0x1.0p-52 - 1.0;
    }
    const double *b = d.b.data(), *a = d.a.data();
    const bool unit = a[0] == 1.0;
    if (sos) {
        const int nsec = N / 2;
        std::vector<double> s3(3 * nsec, 0.0);   /* reference order per section, as sos_step_ref */
        for (int k = 0; k < L; ++k) {
            double xv = x[k];
            for (int q = 0; q < nsec; ++q) {
                double t = xv * b[3 * q];
                t = t + s3[3 * q];
                const double yv = a[3 * q] == 1.0 ? t : t / a[3 * q];
                double u = 0.0 + b[3 * q + 1] * xv;
                u = u - a[3 * q + 1] * yv;
                s3[3 * q] = s3[3 * q + 1] + u;
                u = 0.0 + b[3 * q + 2] * xv;
                u = u - a[3 * q + 2] * yv;
                s3[3 * q + 1] = s3[3 * q + 2] + u;
                xv = yv;
            }
            yref[k] = xv;
        }
    } else {
        std::vector<double> s(S, 0.0);           /* reference order, as iir_step_ref */
        for (int k = 0; k < L; ++k) {
            double t = x[k] * b[0];
            t = t + s[0];
            const double yv = unit ? t : t / a[0];      /* x / 1.0 == x exactly */
            for (int i = 0; i < N; ++i) {
                double u = 0.0 + b[i + 1] * x[k];
                u = u - a[i + 1] * yv;
                s[i] = s[i + 1] + u;
            }
            yref[k] = yv;
        }
    }
    std::vector<double> carry(S, 0.0), v((size_t)R * N), vold((size_t)R * N), s(S);
    for (int seg = 0; seg < 2; ++seg) {
        const double *xs = &x[(size_t)seg * R * T];
        double *ys = &y[(size_t)seg * R * T];
        /* zero-state pass: v_j = sum_n G[n] x[n], n ascending (+ Gc * s[N], the never-written slot) */
        for (int j = 0; j < R; ++j) {
            double *vj = &v[(size_t)j * N];
            for (int i = 0; i < N; ++i) vj[i] = 0.0;
            for (int n = 0; n < T; ++n) {
                const double xv = xs[j * T + n];
                for (int i = 0; i < N; ++i) vj[i] = __builtin_fma(gtab[n * N + i], xv, vj[i]);
            }
            if (carry[N] != 0.0)
                for (int i = 0; i < N; ++i) vj[i] = __builtin_fma(gtab[T * N + i], carry[N], vj[i]);
        }
        /* Kogge-Stone over the rows; levels whose matrix has decayed are skipped as in the kernel */
        for (int lvl = 0; lvl < nlev; ++lvl) {
            const int dist = 1 << lvl;
            if (dist > R) break;
            const double *M = pw + (size_t)lvl * N * N;
            vold = v;
            for (int j = 0; j < R; ++j) {
                const double *u = j >= dist ? &vold[(size_t)(j - dist) * N]
                                            : (!pk.propagators && j == dist - 1 ? carry.data() : nullptr);
                if (!u) continue;
                double *vj = &v[(size_t)j * N];
                for (int i = 0; i < N; ++i) {
                    double acc = vj[i];
                    for (int q = 0; q < N; ++q) acc = __builtin_fma(M[i * N + q], u[q], acc);
                    vj[i] = acc;
                }
            }
        }
        /* true pass: row j from its start state */
        std::vector<double> next(S);
        for (int j = 0; j < R; ++j) {
            if (pk.propagators) {                /* s_j = P_j + A^(T j) carry, P_0 = 0 */
                for (int i = 0; i < N; ++i) {
                    double acc = j ? v[(size_t)(j - 1) * N + i] : 0.0;
                    for (int q = 0; q < N; ++q) acc = __builtin_fma(ttab[(size_t)(i * N + q) * kWave + j], carry[q], acc);
                    s[i] = acc;
                }
            } else {
                for (int i = 0; i < N; ++i) s[i] = j ? v[(size_t)(j - 1) * N + i] : carry[i];
            }
            s[N] = carry[N];
            for (int k = 0; k < T && sos; ++k) {  /* as sos_step_fused */
                double xv = xs[j * T + k];
                for (int q = 0; q < N / 2; ++q) {
                    double yv = __builtin_fma(xv, b[3 * q], s[2 * q]);
                    if (a[3 * q] != 1.0) yv = yv / a[3 * q];
                    s[2 * q] = __builtin_fma(-a[3 * q + 1], yv, __builtin_fma(b[3 * q + 1], xv, s[2 * q + 1]));
                    s[2 * q + 1] = __builtin_fma(-a[3 * q + 2], yv, b[3 * q + 2] * xv);
                    xv = yv;
                }
                ys[j * T + k] = xv;
            }
            for (int k = 0; k < T && !sos; ++k) {        /* as iir_step_fused */
                const double xv = xs[j * T + k];
                double yv = __builtin_fma(xv, b[0], s[0]);
                if (!unit) yv = yv / a[0];
                for (int i = 0; i < N; ++i) s[i] = __builtin_fma(-a[i + 1], yv, __builtin_fma(b[i + 1], xv, s[i + 1]));
                ys[j * T + k] = yv;
            }
            if (j == R - 1) next = s;
        }
        carry = next;
    }
    double peak = 0.0, err = 0.0;
    for (int k = 0; k < L; ++k) {
        peak = std::max(peak, std::fabs(yref[k]));
        const double e = std::fabs(y[k] - yref[k]);
        if (!(e <= err)) err = e;               /* NaN-propagating max */
    }
    if (!(peak > 0.0)) return err == 0.0 ? 0.0 : HUGE_VAL;
    return err / peak;
}

/* hipcc builds host code for baseline x86-64, where __builtin_fma is a libm call (~5 ns): the probe of one
 * design then takes over a millisecond and a bank of 65 536 distinct designs tens of seconds.  Hosts with FMA
 * units (every host of an MI355X) take the instantiation compiled for them; same IEEE results either way. */
template <bool kHw>
static inline __attribute__((always_inline)) double scan_probe_by_order(const Design &d, const ProbeKernel &pk,
                                                                        const double *pw, int nlev, const double *gtab,
                                                                        const double *ttab)
{
    switch (d.fast_n()) {
        case 2: return scan_probe_body<kHw, 2>(d, pk, pw, nlev, gtab, ttab);
        case 4: return scan_probe_body<kHw, 4>(d, pk, pw, nlev, gtab, ttab);
        case 6: return scan_probe_body<kHw, 6>(d, pk, pw, nlev, gtab, ttab);
        case 8: return scan_probe_body<kHw, 8>(d, pk, pw, nlev, gtab, ttab);
        case 12: return scan_probe_body<kHw, 12>(d, pk, pw, nlev, gtab, ttab);
        case 16: return scan_probe_body<kHw, 16>(d, pk, pw, nlev, gtab, ttab);
        default: return scan_probe_body<kHw, 0>(d, pk, pw, nlev, gtab, ttab);
    }
}

__attribute__((target("fma,avx2"))) double scan_probe_hw(const Design &d, const ProbeKernel &pk, const double *pw, int nlev,
                                                         const double *gtab, const double *ttab)
{
    return scan_probe_by_order<true>(d, pk, pw, nlev, gtab, ttab);
}

double scan_probe_sw(const Design &d, const ProbeKernel &pk, const double *pw, int nlev, const double *gtab,
                     const double *ttab)
{
    return scan_probe_by_order<false>(d, pk, pw, nlev, gtab, ttab);
}

double scan_probe(const Design &d, const ProbeKernel &pk, const double *pw, int nlev, const double *gtab,
                  const double *ttab)
{
    static const bool hw = __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2");
    return hw ? scan_probe_hw(d, pk, pw, nlev, gtab, ttab) : scan_probe_sw(d, pk, pw, nlev, gtab, ttab);
}

/* The kernel RTFilterBank_updaten sends a scan-safe design to (see there) */
const ProbeKernel &probe_kernel_of(int order, size_t ndesigns)
{
    if (order > kScan2MaxOrder) return kProbeScan16;
    return ndesigns <= kMaxPipeDesigns ? kProbePipe : kProbeScan2;
}

/* Build every table of one design into `mine` (layout of design_table_doubles) and probe the kernel `pk`. */
struct DesignTables {
    size_t pow = 0, gtab1 = 0, pow2 = 0, gtab = 0, ttab = (size_t)-1;
    int levels1 = 0, levels2 = 0;
    double scan_error = HUGE_VAL;
};

DesignTables build_design_tables(const Design &d, bool with_ttab, const ProbeKernel &pk, std::vector<double> &mine)
{
    DesignTables t;
    mine.clear();
    if (d.sos_fused) {
        /* fused cascade: only the pipelined kernel's tables (layout: A^(32 2^k) | G32 | row propagators) */
        const LinSys sys = sos_system(d);
        t.pow2 = 0;
        sys_powers(sys, kScan2T, kScan2Levels, mine);
        t.levels2 = scan2_levels(mine.data(), sys.N);
        t.gtab = mine.size();
        sys_gtable(sys, kScan2T, mine);
        if (with_ttab) {
            t.ttab = mine.size();
            sys_ttable(sys, kScan2T, mine);
            t.scan_error = scan_probe(d, kProbePipe, &mine[t.pow2], std::min(t.levels2, kScanLevels), &mine[t.gtab],
                                      &mine[t.ttab]);
        }
        return t;
    }
    const int N = d.nb - 1;
    /* A^(64 2^k), k < 6, are levels 1..6 of A^(32 2^k), k < 7 (same squarings, same bits): one chain serves both */
    static_assert(kScanT == 2 * kScan2T && kScan2Levels == kScanLevels + 1, "the two power tables share a chain");
    std::vector<double> chain;
    scan_powers(d, kScan2T, kScan2Levels, chain);
    mine.insert(mine.end(), chain.begin() + (size_t)N * N, chain.end());
    t.levels1 = scan2_levels(mine.data(), N, kScanLevels);
    t.gtab1 = mine.size();
    scan2_gtable(d, kScanT, mine);
    t.pow2 = mine.size();
    mine.insert(mine.end(), chain.begin(), chain.end());
    t.levels2 = scan2_levels(&mine[t.pow2], N);
    t.gtab = mine.size();
    scan2_gtable(d, kScan2T, mine);
    if (N <= kScan2MaxOrder && with_ttab) {
        t.ttab = mine.size();
        pipe_ttable(d, kScan2T, mine);
    }
    if (pk.T == kScanT)
        t.scan_error = scan_probe(d, pk, &mine[t.pow], t.levels1, &mine[t.gtab1], nullptr);
    else if (pk.propagators && t.ttab != (size_t)-1)
        t.scan_error = scan_probe(d, pk, &mine[t.pow2], std::min(t.levels2, kScanLevels), &mine[t.gtab], &mine[t.ttab]);
    else
        t.scan_error = scan_probe(d, kProbeScan2, &mine[t.pow2], t.levels2, &mine[t.gtab], nullptr);
    return t;
}

void free_bank_device(RTFilterBank *b)
{
    for (Group &g : b->groups) {
        pool_free(g.d_chan);
        pool_free(g.d_design);
        pool_free(g.d_state);
        pool_free(g.d_state_alt);
        pool_free(g.d_aux);
        pool_free(g.d_scratch);
        pool_free(g.d_fval);
        pool_free(g.d_init);
    }
    pool_free(b->d_coef_off);
    pool_free(b->d_pow_off);
    pool_free(b->d_pow2_off);
    pool_free(b->d_levels2);
    pool_free(b->d_gtab_off);
    pool_free(b->d_levels1);
    pool_free(b->d_gtab1_off);
    pool_free(b->d_ttab_off);
    pool_free(b->d_nb);
    pool_free(b->d_na);
    pool_free(b->d_kind);
    pool_free(b->d_coefs);
    pool_free(b->d_pows);
}

/* From this many channels in a group the sequential (lane = channel) kernel fills the GPU and AUTO
 * uses it; below, the time-parallel scan kernels. */
constexpr size_t kAutoSeqMinChannels = 65536;
constexpr size_t kSeqOnlyChannels = 2 * kAutoSeqMinChannels;   /* orders <= 8: sequential in every mode from here */

/* Doubles of scan tables one design owns in `pows` (layout: A^(64 2^k) | G64 | A^(32 2^k) | G32 | row propagators). */
size_t design_table_doubles(const Design &d, bool with_ttab)
{
    if (!d.has_tables) return 0;
    if (d.sos_fused) {
        const size_t N = (size_t)d.fast_n();
        return (size_t)kScan2Levels * N * N + ((size_t)kScan2T + 1) * N + (with_ttab ? (size_t)kWave * N * N : 0);
    }
    const size_t N = (size_t)d.nb - 1;
    size_t n = (size_t)kScanLevels * N * N + ((size_t)kScanT + 1) * N + (size_t)kScan2Levels * N * N +
               ((size_t)kScan2T + 1) * N;
    if (N <= (size_t)kScan2MaxOrder && with_ttab) n += (size_t)kWave * N * N;
    return n;
}

/* Common tail of both constructors: designs + design_of are filled in;
 * init_state (nch x state_ld, may be empty = zeros), init_fval, init_init
 * give the starting per-channel fields. */
int finish_bank(RTFilterBank *bk, const std::vector<double> &init_state,
                const std::vector<double> &init_fval, const std::vector<int32_t> &init_init)
{
    const int device = bk->device;
    DeviceScope ds(device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    {
        int rc0 = static_scratch_of(device, &bk->d_static_scratch);
        if (rc0) return rc0;
    }

    /* design tables.  Offsets are 32-bit in the kernels' view (GroupView): sizes are laid out first and
     * checked, then the tables of the designs are built in parallel, each into its own slice. */
    std::vector<int32_t> coef_off, pow_off, pow2_off, levels2, gtab_off, ttab_off, levels1, gtab1_off, nb, na, kind;
    std::vector<double> coefs, pows;
    size_t ld = 1;
    /* RTFB_KIND_SOS designs: every section becomes an IIR design of its own (nb = na = 3) that the ordinary
     * kernels run; the channel's sections are chained through launch groups of increasing `stage`. */
    {
        const size_t nd0 = bk->designs.size();
        /* Cascades of up to kMaxFusedSections sections with a0 == 1 everywhere are FUSED: one launch group, all
         * sections per sample, one pass over the block (needs the pipelined kernel's per-design row propagators:
         * banks of <= kMaxPipeDesigns designs).  The others keep one pass per section. */
        for (size_t i = 0; i < nd0; ++i) {
            Design &d = bk->designs[i];
            if (d.kind != RTFB_KIND_SOS) continue;
            bool unit = true;
            for (int k = 0; k < d.nb / 3; ++k) unit = unit && d.a[3 * k] == 1.0;
            d.sos_fused = unit && d.nb / 3 <= kMaxFusedSections && nd0 <= kMaxPipeDesigns;
        }
        for (size_t i = 0; i < nd0; ++i) {
            if (bk->designs[i].kind != RTFB_KIND_SOS || bk->designs[i].sos_fused) continue;
            const int nsec = bk->designs[i].nb / 3;
            for (int k = 0; k < nsec; ++k) {
                Design sec;
                sec.kind = RTFB_KIND_IIR;
                sec.nb = sec.na = 3;
                sec.b.assign(bk->designs[i].b.begin() + 3 * k, bk->designs[i].b.begin() + 3 * k + 3);
                sec.a.assign(bk->designs[i].a.begin() + 3 * k, bk->designs[i].a.begin() + 3 * k + 3);
                bk->designs[i].section_design.push_back((int)bk->designs.size());
                bk->designs.push_back(std::move(sec));
            }
        }
    }
    /* (channel, stage) -> the design a launch group runs there */
    auto staged = [](const Design &d) { return d.kind == RTFB_KIND_SOS && !d.sos_fused; };
    auto stages_of = [&](size_t c) { const Design &d = bk->designs[bk->design_of[c]];
                                     return staged(d) ? (int)d.section_design.size() : 1; };
    auto stage_design = [&](size_t c, int stage) { const Design &d = bk->designs[bk->design_of[c]];
                                                   return staged(d) ? d.section_design[stage] : bk->design_of[c]; };
    {
        /* Designs whose launch group can never take a scan kernel carry no scan tables and skip the
         * conditioning probe: orders <= 8 in a class of >= 2 * kAutoSeqMinChannels channels run the
         * lane-per-channel kernel in every mode (RTFilterBank_updaten). */
        std::map<int, size_t> class_channels;                    /* fast-IIR class (per stage) -> channels */
        auto class_of = [](const Design &d, int stage) {
            return d.sos_fused ? 100000 + d.fast_n() : 64 * stage + 2 * (d.nb - 1) + (d.a[0] == 1.0 ? 1 : 0);
        };
        std::vector<int> design_stage(bk->designs.size(), 0);
        for (size_t c = 0; c < bk->nch; ++c)
            for (int k = 0; k < stages_of(c); ++k) {
                const int di = stage_design(c, k);
                const Design &d = bk->designs[di];
                design_stage[di] = k;
                if (d.fast()) class_channels[class_of(d, k)]++;
            }
        const bool with_ttab = bk->designs.size() <= kMaxPipeDesigns;
        size_t ncoef = 0, npow = 0;
        for (size_t di = 0; di < bk->designs.size(); ++di) {
            Design &d = bk->designs[di];
            d.has_tables = false;
            if (d.fast()) {
                const int N = d.fast_n();
                const auto it = class_channels.find(class_of(d, design_stage[di]));
                const size_t width = it == class_channels.end() ? 0 : it->second;
                d.has_tables = width > 0 && !(N <= kScan2MaxOrder && width >= kSeqOnlyChannels);
            }
            ncoef += (size_t)d.nb + (size_t)d.na;
            npow += design_table_doubles(d, with_ttab);
        }
        if (ncoef > (size_t)INT32_MAX || npow > (size_t)INT32_MAX)
            return fail(RTFB_ESIZE, "the bank's design tables exceed 2^31 doubles (" + std::to_string(npow) +
                                        " of scan tables for " + std::to_string(bk->designs.size()) +
                                        " designs); split the bank");
        pows.assign(std::max<size_t>(npow, 1), 0.0);
        coefs.reserve(ncoef);
        size_t poff = 0;
        for (Design &d : bk->designs) {
            d.coef_off = (int)coefs.size();
            coefs.insert(coefs.end(), d.b.begin(), d.b.end());
            coefs.insert(coefs.end(), d.a.begin(), d.a.end());
            d.pow_off = (int)poff;
            poff += design_table_doubles(d, with_ttab);
        }
        /* build: one design at a time per worker, no shared state beyond the disjoint slices */
        std::atomic<size_t> next{0};
        const size_t nd = bk->designs.size();
        auto worker = [&]() {
            std::vector<double> mine;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= nd) break;
                Design &d = bk->designs[i];
                if (!d.has_tables) continue;
                const size_t base = (size_t)d.pow_off;
                const DesignTables t = build_design_tables(d, with_ttab, probe_kernel_of(d.fast_n(), nd), mine);
                d.scan_error = t.scan_error;
                d.levels1 = t.levels1;
                d.levels2 = t.levels2;
                d.gtab1_off = (int)(base + t.gtab1);
                d.pow2_off = (int)(base + t.pow2);
                d.gtab_off = (int)(base + t.gtab);
                if (t.ttab != (size_t)-1) d.ttab_off = (int)(base + t.ttab);
                std::copy(mine.begin(), mine.end(), pows.begin() + (long)base);
            }
        };
        size_t nthreads = std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16,
                                            (nd + 7) / 8});
        if (nthreads <= 1) worker();
        else {
            std::vector<std::thread> pool_threads;
            for (size_t t = 0; t < nthreads; ++t) pool_threads.emplace_back(worker);
            for (std::thread &t : pool_threads) t.join();
        }
    }
    for (Design &d : bk->designs) {
        gtab_off.push_back(d.gtab_off);
        ttab_off.push_back(d.ttab_off);
        levels1.push_back(d.levels1);
        gtab1_off.push_back(d.gtab1_off);
        pow2_off.push_back(d.pow2_off);
        levels2.push_back(d.levels2);
        coef_off.push_back(d.coef_off);
        pow_off.push_back(d.pow_off);
        nb.push_back(d.nb);
        na.push_back(d.na);
        kind.push_back(d.kind);
        ld = std::max(ld, (size_t)d.state_slots());
    }
    bk->state_ld = ld;

    /* groups */
    std::map<std::tuple<int, int, int>, size_t> index;          /* (stage, class, subkey) -> group */
    for (size_t c = 0; c < bk->nch; ++c) {
        for (int stage = 0; stage < stages_of(c); ++stage) {
            const int di = stage_design(c, stage);
            const Design &d = bk->designs[di];
            int cls, sub = 0;
            if (d.sos_fused) {
                cls = kIirFast;
                sub = 1024 + 4 * d.fast_n() + 2 + (d.scan_safe() ? 1 : 0);       /* a0 == 1 in every section */
            } else if (d.kind != RTFB_KIND_IIR) cls = kFir;
            else if (d.fast_iir()) {
                cls = kIirFast;
                sub = 4 * (d.nb - 1) + (d.a[0] == 1.0 ? 2 : 0) + (d.scan_safe() ? 1 : 0);
            } else cls = kIirGeneric;
            const auto key = std::make_tuple(stage, cls, sub);
            auto it = index.find(key);
            if (it == index.end()) {
                Group g;
                g.cls = (GroupClass)cls;
                g.stage = stage;
                g.order = (sub & 1023) / 4;
                g.sos_sections = sub >= 1024 ? g.order / 2 : 0;
                g.unit_a0 = (sub & 2) != 0;
                g.scan_safe = (sub & 1) != 0;
                it = index.emplace(key, bk->groups.size()).first;
                bk->groups.push_back(g);
            }
            Group &g = bk->groups[it->second];
            g.chans.push_back((int32_t)c);
            g.chan_design.push_back(di);
            g.state_ld = std::max(g.state_ld, d.state_slots());
            g.levels2_max = std::max(g.levels2_max, d.levels2);
            if (d.fast() && !d.has_tables) g.has_tables = false;
            if (g.cls == kIirFast && g.even_b) {
                if (d.sos_fused) g.even_b = false;
                for (size_t i = 1; i < d.b.size() && g.even_b; i += 2)
                    if (d.b[i] != 0.0) g.even_b = false;
            }
            if (d.kind != RTFB_KIND_IIR && !d.sos_fused) g.kmax = std::max(g.kmax, d.nb);
            if (d.kind == RTFB_KIND_FIR_PARTIAL) g.has_partial = true;
            if (g.single_design == -2) g.single_design = di;
            else if (g.single_design != di) g.single_design = -1;
        }
    }
    /* launch order = stage order (the map iterates stages ascending; groups were appended in channel order) */
    std::stable_sort(bk->groups.begin(), bk->groups.end(),
                     [](const Group &p, const Group &q) { return p.stage < q.stage; });

    int rc;
    if ((rc = upload(&bk->d_coef_off, coef_off, device))) return rc;
    if ((rc = upload(&bk->d_pow_off, pow_off, device))) return rc;
    if ((rc = upload(&bk->d_pow2_off, pow2_off, device))) return rc;
    if ((rc = upload(&bk->d_levels2, levels2, device))) return rc;
    if ((rc = upload(&bk->d_gtab_off, gtab_off, device))) return rc;
    if ((rc = upload(&bk->d_levels1, levels1, device))) return rc;
    if ((rc = upload(&bk->d_gtab1_off, gtab1_off, device))) return rc;
    if ((rc = upload(&bk->d_ttab_off, ttab_off, device))) return rc;
    if ((rc = upload(&bk->d_nb, nb, device))) return rc;
    if ((rc = upload(&bk->d_na, na, device))) return rc;
    if ((rc = upload(&bk->d_kind, kind, device))) return rc;
    if ((rc = upload(&bk->d_coefs, coefs, device))) return rc;
    if ((rc = upload(&bk->d_pows, pows, device))) return rc;

    for (Group &g : bk->groups) {
        const size_t cnt = g.chans.size();
        g.contiguous = (size_t)(g.chans.back() - g.chans.front()) + 1 == cnt;
        std::vector<int32_t> dsn(cnt), ini(cnt);
        std::vector<double> st(cnt * (size_t)g.state_ld, 0.0), fv(cnt);
        for (size_t i = 0; i < cnt; ++i) {
            const size_t c = (size_t)g.chans[i];
            dsn[i] = g.chan_design[i];
            fv[i] = init_fval.empty() ? 0.0 : init_fval[c];
            ini[i] = init_init.empty() ? FILTER_RESET : init_init[c];
            if (!init_state.empty()) {
                /* a section of an SOS channel owns slots [3 stage, 3 stage + 3) of the channel's state row */
                const bool sos = staged(bk->designs[bk->design_of[c]]);
                const int slots = bk->designs[dsn[i]].state_slots();
                const size_t off = sos ? (size_t)3 * g.stage : 0;
                for (int k = 0; k < slots; ++k) st[i * g.state_ld + k] = init_state[c * ld + off + k];
            }
        }
        if (!g.contiguous && (rc = upload(&g.d_chan, g.chans, device))) return rc;
        if ((rc = upload(&g.d_design, dsn, device))) return rc;
        if ((rc = upload(&g.d_state, st, device))) return rc;
        if (g.cls == kFir) {
            if ((rc = upload(&g.d_state_alt, st, device))) return rc;
            std::vector<int32_t> aux(4 * cnt + 2, 0);      /* + a 64-bit word per channel and one int (fir_kernels.hip: nf_mark) */
            if ((rc = upload(&g.d_aux, aux, device))) return rc;
        }
        if (g.cls == kIirFast && std::getenv("DSP_AMD_SCAN_STAMPS")) {   /* diagnostics only */
            std::vector<int32_t> dbg(64 * cnt, 0);
            if ((rc = upload(&g.d_aux, dbg, device))) return rc;
        }
        if ((rc = upload(&g.d_fval, fv, device))) return rc;
        if ((rc = upload(&g.d_init, ini, device))) return rc;
    }
    return RTFB_OK;
}

GroupView view_of(const RTFilterBank *bk, const Group &g)
{
    GroupView v;
    v.chan = g.contiguous ? nullptr : g.d_chan;
    v.chan0 = g.chans.front();
    v.count = (int32_t)g.chans.size();
    v.design = g.d_design;
    v.state = g.d_state;
    v.state_out = g.d_state_alt;
    v.state_ld = g.state_ld;
    v.aux = g.d_aux;
    v.kmax = g.kmax;
    v.unit_a0 = g.unit_a0 ? 1 : 0;
    v.even_b = (g.cls == kIirFast && g.even_b) ? 1 : 0;
    v.fir_mfma = 0;
    v.has_partial = g.has_partial ? 1 : 0;
    v.single_design = g.single_design;
    v.fval = g.d_fval;
    v.init = g.d_init;
    v.coef_off = bk->d_coef_off;
    v.pow_off = bk->d_pow_off;
    v.pow2_off = bk->d_pow2_off;
    v.levels2 = bk->d_levels2;
    v.gtab_off = bk->d_gtab_off;
    v.levels1 = bk->d_levels1;
    v.gtab1_off = bk->d_gtab1_off;
    v.ttab_off = bk->d_ttab_off;
    v.levels2_max = g.levels2_max;
    v.nb = bk->d_nb;
    v.na = bk->d_na;
    v.kind = bk->d_kind;
    v.coefs = bk->d_coefs;
    v.pows = bk->d_pows;
    return v;
}

/* ---- chained block updates (RTFilterBank_updaten_chained) ---------------------------------------------------- */

void chain_release(RTFilterBank *bk)
{
    ChainState &c = bk->chain;
    for (int k = 0; k < 2; ++k)
        if (c.s[k]) {
            (void)hipStreamSynchronize(c.s[k]);
            (void)hipStreamDestroy(c.s[k]);
        }
    for (int k = 0; k < kChainRing; ++k) {
        if (c.done[k]) (void)hipEventDestroy(c.done[k]);
        if (c.in_ready[k]) (void)hipEventDestroy(c.in_ready[k]);
    }
    pool_free(c.d_queue);
    pool_free(c.d_counters);
    pool_free(c.d_error);
    pool_free(c.d_records);
    if (c.h_error) (void)hipHostFree(c.h_error);
    c = ChainState();
}

/* The one launch group a chain can run on, or nullptr: a single group of orders <= 8 that the pipelined kernel
 * serves (tables present, design admitted by the probe or RTFB_MODE_SCAN), no wider than one resident round. */
Group *chain_group(RTFilterBank *bk)
{
    if (bk->groups.size() != 1) return nullptr;
    Group &g = bk->groups[0];
    if (g.cls != kIirFast || g.stage != 0 || g.order < 1 || g.order > kScan2MaxOrder || !g.has_tables || g.sos_sections)
        return nullptr;
    if (bk->designs.size() > kMaxPipeDesigns) return nullptr;
    if (bk->mode == RTFB_MODE_SEQUENTIAL || (bk->mode == RTFB_MODE_AUTO && !g.scan_safe)) return nullptr;
    return &g;
}

/* streams, events, pinned error word (also what a call that falls back to stream order hands out as done_event) */
int chain_prepare_events(RTFilterBank *bk)
{
    ChainState &c = bk->chain;
    if (c.have_events) return RTFB_OK;
    int rc = RTFB_OK;
    auto hipok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) rc = fail_hip(e, what);
        return e == hipSuccess;
    };
    for (int k = 0; k < 2 && !rc; ++k) hipok(hipStreamCreateWithFlags(&c.s[k], hipStreamNonBlocking), "hipStreamCreate");
    for (int k = 0; k < kChainRing && !rc; ++k)
        hipok(hipEventCreateWithFlags(&c.done[k], hipEventDisableTiming), "hipEventCreate") &&
            hipok(hipEventCreateWithFlags(&c.in_ready[k], hipEventDisableTiming), "hipEventCreate");
    if (!rc) hipok(hipHostMalloc(reinterpret_cast<void **>(&c.h_error), 64, hipHostMallocDefault), "hipHostMalloc");
    if (rc) {
        const std::string keep = t_last_error;
        chain_release(bk);
        t_last_error = keep;
        return rc;
    }
    *c.h_error = 0;
    c.have_events = true;
    return RTFB_OK;
}

int chain_prepare(RTFilterBank *bk, const Group &g)
{
    ChainState &c = bk->chain;
    if (c.ready) return RTFB_OK;
    int rc = chain_prepare_events(bk);
    if (rc) return rc;
    const int device = bk->device;
    const size_t cnt = g.chans.size();
    auto hipok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) rc = fail_hip(e, what);
        return e == hipSuccess;
    };
    void *p = nullptr;
    const size_t qbytes = (size_t)kChainRing * cnt * sizeof(unsigned long long);
    const size_t cbytes = (size_t)2 * kChainRing * 16 * sizeof(unsigned long long);
    const size_t rbytes = cnt * (size_t)(2 * (g.order + 2)) * sizeof(double);
    if (!rc && !(rc = pool_alloc(&p, qbytes, device))) c.d_queue = static_cast<unsigned long long *>(p);
    if (!rc && !(rc = pool_alloc(&p, cbytes, device))) c.d_counters = static_cast<unsigned long long *>(p);
    if (!rc && !(rc = pool_alloc(&p, 256, device))) c.d_error = static_cast<unsigned int *>(p);
    if (!rc && !(rc = pool_alloc(&p, rbytes, device))) c.d_records = static_cast<double *>(p);
    if (!rc) {
        hipok(hipMemset(c.d_queue, 0, qbytes), "hipMemset") && hipok(hipMemset(c.d_counters, 0, cbytes), "hipMemset") &&
            hipok(hipMemset(c.d_error, 0, 256), "hipMemset") && hipok(hipMemset(c.d_records, 0, rbytes), "hipMemset") &&
            hipok(hipDeviceSynchronize(), "hipDeviceSynchronize");
    }
    if (rc) {
        const std::string keep = t_last_error;
        chain_release(bk);
        t_last_error = keep;
        return rc;
    }
    c.count = (int)cnt;
    c.order = g.order;
    c.ready = true;
    return RTFB_OK;
}

/* End the open chain: `s` waits for every link, the carry records go back into the group's plain arrays, the
 * error word is copied to the host.  Enqueues only.  Every other entry point of the bank calls this first. */
int chain_close(RTFilterBank *bk, hipStream_t s)
{
    ChainState &c = bk->chain;
    if (c.h_error && *c.h_error)
        return fail(RTFB_EDEVICE, "RTFilterBank_updaten_chained: a bounded wait between chained launches ran out (code " +
                                      std::to_string(*c.h_error) + "); the bank's state is undefined, rebuild it");
    if (!c.open) return RTFB_OK;
    DeviceScope ds(bk->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    const uint64_t last = c.seq - 1;
    HIP_TRY(hipStreamWaitEvent(s, c.done[last % kChainRing], 0));
    if (c.links >= 2) HIP_TRY(hipStreamWaitEvent(s, c.done[(last - 1) % kChainRing], 0));
    Group &g = bk->groups[0];
    HIP_TRY(launch_chain_unpack(c.d_records, g.d_state, g.d_fval, g.d_init, c.count, c.order, g.state_ld, s));
    HIP_TRY(hipMemcpyAsync(c.h_error, c.d_error, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    c.open = false;
    c.links = 0;
    return RTFB_OK;
}

int validate_design(const RTFilterDesign &d)
{
    if (d.kind != RTFB_KIND_IIR && d.kind != RTFB_KIND_FIR && d.kind != RTFB_KIND_FIR_PARTIAL &&
        d.kind != RTFB_KIND_SOS)
        return fail(RTFB_EINVAL, "design.kind is not an RTFB_KIND_* value");
    if (!d.b || d.nb == 0) return fail(RTFB_EINVAL, "design needs at least one b tap");
    if (d.kind == RTFB_KIND_SOS) {
        if (!d.a || d.na != d.nb || d.nb % 3 || d.nb / 3 > (size_t)DSP_MAX_SECTIONS)
            return fail(RTFB_EINVAL, "SOS design: b and a hold three taps per section, at most 64 sections");
        return RTFB_OK;
    }
    if (d.kind == RTFB_KIND_IIR) {
        if (!d.a || d.na == 0) return fail(RTFB_EINVAL, "IIR design needs at least one a tap");
        if (d.nb > (size_t)kMaxGenericTaps || d.na > (size_t)kMaxGenericTaps)
            return fail(RTFB_ESIZE, "IIR design exceeds 64 taps");
    } else if (d.nb > (size_t)kMaxFirTaps) {
        return fail(RTFB_ESIZE, "FIR design exceeds 4096 taps");
    }
    return RTFB_OK;
}

}  // namespace

/* ======================================================================== */
/* C ABI                                                                      */

extern "C" {

const char *RTFilterBank_last_error(void) { return t_last_error.c_str(); }

int dsp_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

int dsp_pool_alloc(void **dev_ptr, size_t bytes, int device) { return pool_alloc(dev_ptr, bytes, device); }

/* hipFree semantics: the device is synchronised first, so no kernel on any stream can still be using the block
 * when the next dsp_pool_alloc hands it out.  Real-time callers use dsp_pool_free_async. */
int dsp_pool_free(void *dev_ptr)
{
    if (!dev_ptr) return RTFB_OK;
    int device = -1;
    {
        Pool &p = pool();
        std::lock_guard<std::mutex> lk(p.mu);
        auto it = p.live.find(dev_ptr);
        if (it == p.live.end()) return fail(RTFB_EINVAL, "dsp_pool_free: pointer not from this pool");
        device = it->second.device;
    }
    {
        DeviceScope ds(device);
        if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
        HIP_TRY(hipDeviceSynchronize());
    }
    return pool_free(dev_ptr);
}

int dsp_pool_free_async(void *dev_ptr, void *hip_stream)
{
    return pool_free(dev_ptr, true, static_cast<hipStream_t>(hip_stream));
}

int dsp_pool_trim(int device)
{
    Pool &p = pool();
    std::vector<void *> victims;
    {
        std::lock_guard<std::mutex> lk(p.mu);
        for (auto &kv : p.cache) {
            if (kv.first.first != device) continue;
            for (const Pool::Cached &c : kv.second) {
                victims.push_back(c.ptr);
                if (c.ready) (void)hipEventDestroy(c.ready);      /* hipFree below synchronises by itself */
            }
            p.cached[device] -= kv.first.second * kv.second.size();
            kv.second.clear();
        }
    }
    DeviceScope ds(device);
    for (void *ptr : victims) (void)hipFree(ptr);
    return RTFB_OK;
}

size_t dsp_pool_bytes_in_use(int device)
{
    Pool &p = pool();
    std::lock_guard<std::mutex> lk(p.mu);
    return p.in_use[device];
}

size_t dsp_pool_bytes_cached(int device)
{
    Pool &p = pool();
    std::lock_guard<std::mutex> lk(p.mu);
    return p.cached[device];
}

int dsp_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, int device)
{
    DeviceScope ds(device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return RTFB_OK;
}

int dsp_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, int device)
{
    DeviceScope ds(device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return RTFB_OK;
}

int dsp_stream_synchronize(void *hip_stream, int device)
{
    DeviceScope ds(device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)));
    return RTFB_OK;
}

int dsp_synth_fill(double *dst, size_t ld, size_t nch, size_t n, uint64_t seed, uint64_t chan0,
                   uint64_t n0, void *hip_stream)
{
    if (!dst && nch * n) return fail(RTFB_EINVAL, "dsp_synth_fill: NULL destination");
    if (ld < n) return fail(RTFB_EINVAL, "dsp_synth_fill: ld < n");
    HIP_TRY(launch_synth(dst, ld, nch, n, seed, chan0, n0, static_cast<hipStream_t>(hip_stream)));
    return RTFB_OK;
}

int dsp_scan_probe(const RTFilterDesign *design, double *rel_error)
{
    if (!design || !rel_error) return fail(RTFB_EINVAL, "dsp_scan_probe: NULL argument");
    int rc = validate_design(*design);
    if (rc) return rc;
    Design d;
    d.kind = design->kind;
    d.nb = (int)design->nb;
    d.na = (d.kind == RTFB_KIND_IIR || d.kind == RTFB_KIND_SOS) ? (int)design->na : 0;
    d.b.assign(design->b, design->b + d.nb);
    if (d.na) d.a.assign(design->a, design->a + d.na);
    if (!d.fast_iir()) return fail(RTFB_EUNSUPPORTED, "the scan kernel takes IIR designs with na == nb <= 17");
    std::vector<double> mine;
    *rel_error = build_design_tables(d, true, probe_kernel_of(d.nb - 1, 1), mine).scan_error;
    return RTFB_OK;
}

int dsp_scan_probe_kernel(const RTFilterDesign *design, int kernel, double *rel_error)
{
    if (!design || !rel_error) return fail(RTFB_EINVAL, "dsp_scan_probe_kernel: NULL argument");
    int rc = validate_design(*design);
    if (rc) return rc;
    Design d;
    d.kind = design->kind;
    d.nb = (int)design->nb;
    d.na = d.kind == RTFB_KIND_IIR ? (int)design->na : 0;
    d.b.assign(design->b, design->b + d.nb);
    if (d.na) d.a.assign(design->a, design->a + d.na);
    if (!d.fast_iir()) return fail(RTFB_EUNSUPPORTED, "the scan kernels take IIR designs with na == nb <= 17");
    const int N = d.nb - 1;
    const ProbeKernel *pk = nullptr;
    if (kernel == RTFB_SCAN_KERNEL_PIPE && N <= kScan2MaxOrder) pk = &kProbePipe;
    else if (kernel == RTFB_SCAN_KERNEL_SCAN2 && N <= kScan2MaxOrder) pk = &kProbeScan2;
    else if (kernel == RTFB_SCAN_KERNEL_SCAN16 && N > kScan2MaxOrder) pk = &kProbeScan16;
    if (!pk) return fail(RTFB_EINVAL, "dsp_scan_probe_kernel: that kernel does not serve this order");
    std::vector<double> mine;
    *rel_error = build_design_tables(d, true, *pk, mine).scan_error;
    return RTFB_OK;
}

/* Host arithmetic only (no device needed): doubles of scan tables a bank of these designs would carry if every
 * design kept its tables, and whether that fits the kernels' 32-bit table offsets (RTFB_ESIZE if not) -- the check
 * RTFilterBank_from_designs / _from_filters apply before building anything. */
int dsp_bank_table_doubles(const RTFilterDesign *designs, size_t ndesign, size_t *doubles)
{
    if (!designs || !ndesign || !doubles) return fail(RTFB_EINVAL, "dsp_bank_table_doubles: NULL or empty argument");
    size_t total = 0, ncoef = 0;
    for (size_t i = 0; i < ndesign; ++i) {
        int rc = validate_design(designs[i]);
        if (rc) return rc;
        Design d;
        d.kind = designs[i].kind;
        d.nb = (int)designs[i].nb;
        d.na = (d.kind == RTFB_KIND_IIR || d.kind == RTFB_KIND_SOS) ? (int)designs[i].na : 0;
        d.has_tables = d.fast_iir();
        total += design_table_doubles(d, ndesign <= kMaxPipeDesigns);
        ncoef += (size_t)d.nb + (size_t)d.na;
    }
    *doubles = total;
    if (total > (size_t)INT32_MAX || ncoef > (size_t)INT32_MAX)
        return fail(RTFB_ESIZE, "the bank's design tables exceed 2^31 doubles; split the bank");
    return RTFB_OK;
}

int RTFilterBank_from_designs(RTFilterBank **bank, const RTFilterDesign *designs, size_t ndesign,
                              const int32_t *design_of_channel, size_t nch, int device)
{
    if (!bank || !designs || !design_of_channel || !ndesign || !nch)
        return fail(RTFB_EINVAL, "RTFilterBank_from_designs: NULL or empty argument");
    *bank = nullptr;
    if (nch > (size_t)INT32_MAX) return fail(RTFB_ESIZE, "too many channels");
    int rc = check_device(device);
    if (rc) return rc;
    RTFilterBank *bk = new RTFilterBank;
    bk->device = device;
    bk->nch = nch;
    for (size_t i = 0; i < ndesign; ++i) {
        if ((rc = validate_design(designs[i]))) { delete bk; return rc; }
        Design d;
        d.kind = designs[i].kind;
        d.nb = (int)designs[i].nb;
        d.na = (d.kind == RTFB_KIND_IIR || d.kind == RTFB_KIND_SOS) ? (int)designs[i].na : 0;
        d.b.assign(designs[i].b, designs[i].b + d.nb);
        if (d.na) d.a.assign(designs[i].a, designs[i].a + d.na);
        bk->designs.push_back(std::move(d));
    }
    bk->design_of.assign(design_of_channel, design_of_channel + nch);
    for (int32_t d : bk->design_of)
        if (d < 0 || (size_t)d >= ndesign) {
            delete bk;
            return fail(RTFB_EINVAL, "design_of_channel entry out of range");
        }
    rc = finish_bank(bk, {}, {}, {});
    if (rc) {
        free_bank_device(bk);
        delete bk;
        return rc;
    }
    *bank = bk;
    return RTFB_OK;
}

int RTFilterBank_from_filters(RTFilterBank **bank, RTFilter *const *filters, size_t nch, int device)
{
    if (!bank || !filters || !nch)
        return fail(RTFB_EINVAL, "RTFilterBank_from_filters: NULL or empty argument");
    *bank = nullptr;
    if (nch > (size_t)INT32_MAX) return fail(RTFB_ESIZE, "too many channels");
    int rc = check_device(device);
    if (rc) return rc;

    RTFilterBank *bk = new RTFilterBank;
    bk->device = device;
    bk->nch = nch;
    bk->design_of.resize(nch);
    bk->host_filters.assign(filters, filters + nch);
    std::map<std::string, int> seen;            /* coefficient bytes -> design index */
    std::vector<const double *> st_ptr(nch);
    std::deque<std::vector<double>> sos_taps;   /* unpacked sections of FILTER_SOS filters (stable addresses) */
    size_t ld = 1;
    for (size_t c = 0; c < nch; ++c) {
        RTFilter *f = filters[c];
        if (!f) { delete bk; return fail(RTFB_EINVAL, "NULL filter in array"); }
        RTFilterDesign rd;
        std::memset(&rd, 0, sizeof rd);
        /* acceleration rule: callbacks must be the library's own, by pointer identity */
        if (f->update == RTIIRFilter_update && f->initialize == RTIIRFilter_stable_init) {
            RTIIRFilter *q = reinterpret_cast<RTIIRFilter *>(f);
            rd.kind = RTFB_KIND_IIR;
            rd.nb = q->ifb.fb.nb;
            rd.na = q->ifb.na;
            rd.b = q->ifb.fb.b;
            rd.a = q->ifb.fb.b + q->ifb.fb.nb;
            st_ptr[c] = q->state;
        } else if (f->update == RTIIRFilter_sos_update && f->initialize == RTIIRFilter_sos_stable_init) {
            RTIIRFilter *q = reinterpret_cast<RTIIRFilter *>(f);
            RTFilterSection sec[DSP_MAX_SECTIONS];
            size_t ns = 0;
            if (RTIIRFilter_sos_sections(q, sec, DSP_MAX_SECTIONS, &ns) || !ns) {
                delete bk;
                return fail(RTFB_EINVAL, "FILTER_SOS filter without sections");
            }
            sos_taps.emplace_back(6 * ns);
            std::vector<double> &t = sos_taps.back();       /* b of all sections, then a of all sections */
            for (size_t k = 0; k < ns; ++k)
                for (int j = 0; j < 3; ++j) {
                    t[3 * k + j] = sec[k].b[j];
                    t[3 * ns + 3 * k + j] = sec[k].a[j];
                }
            rd.kind = RTFB_KIND_SOS;
            rd.nb = rd.na = 3 * ns;
            rd.b = t.data();
            rd.a = t.data() + 3 * ns;
            st_ptr[c] = q->state;
        } else if (f->update == RTFIRFilter_update &&
                   (f->initialize == RTFIRFilter_stable_init ||
                    f->initialize == RTFIRFilter_partial_init)) {
            RTFIRFilter *q = reinterpret_cast<RTFIRFilter *>(f);
            rd.kind = f->initialize == RTFIRFilter_partial_init ? RTFB_KIND_FIR_PARTIAL : RTFB_KIND_FIR;
            rd.nb = q->fb.nb;
            rd.b = q->fb.b;
            st_ptr[c] = q->state;
        } else {
            delete bk;
            return fail(RTFB_EUNSUPPORTED,
                        "filter carries update/initialize callbacks that are not this library's");
        }
        if ((rc = validate_design(rd)) || !st_ptr[c]) {
            delete bk;
            return rc ? rc : fail(RTFB_EINVAL, "filter has no state buffer");
        }
        std::string key(reinterpret_cast<const char *>(&rd.kind), sizeof rd.kind);
        key.append(reinterpret_cast<const char *>(&rd.nb), sizeof rd.nb);
        key.append(reinterpret_cast<const char *>(&rd.na), sizeof rd.na);
        key.append(reinterpret_cast<const char *>(rd.b), rd.nb * sizeof(double));
        if (rd.na) key.append(reinterpret_cast<const char *>(rd.a), rd.na * sizeof(double));
        auto it = seen.find(key);
        if (it == seen.end()) {
            Design d;
            d.kind = rd.kind;
            d.nb = (int)rd.nb;
            d.na = (int)rd.na;
            d.b.assign(rd.b, rd.b + rd.nb);
            if (rd.na) d.a.assign(rd.a, rd.a + rd.na);
            it = seen.emplace(key, (int)bk->designs.size()).first;
            bk->designs.push_back(std::move(d));
        }
        bk->design_of[c] = it->second;
        ld = std::max(ld, (size_t)bk->designs[it->second].state_slots());
    }
    std::vector<double> st(nch * ld, 0.0), fv(nch);
    std::vector<int32_t> ini(nch);
    for (size_t c = 0; c < nch; ++c) {
        const Design &d = bk->designs[bk->design_of[c]];
        const int slots = d.state_slots();
        if (d.kind == RTFB_KIND_SOS) {              /* host: two doubles per section; here three (the last is 0) */
            for (int k = 0; k < slots / 3; ++k) {
                st[c * ld + 3 * k] = st_ptr[c][2 * k];
                st[c * ld + 3 * k + 1] = st_ptr[c][2 * k + 1];
            }
        } else
            std::memcpy(&st[c * ld], st_ptr[c], slots * sizeof(double));
        fv[c] = filters[c]->filtered_value;
        ini[c] = filters[c]->initialized;
    }
    rc = finish_bank(bk, st, fv, ini);
    if (rc) {
        free_bank_device(bk);
        delete bk;
        return rc;
    }
    *bank = bk;
    return RTFB_OK;
}

void RTFilterBank_destroy(RTFilterBank *bank)
{
    if (!bank) return;
    {
        DeviceScope ds(bank->device);
        (void)hipDeviceSynchronize();
    }
    host_pipe_release(bank->host);
    chain_release(bank);
    free_bank_device(bank);
    delete bank;
}

size_t RTFilterBank_channels(const RTFilterBank *bank) { return bank ? bank->nch : 0; }
size_t RTFilterBank_state_ld(const RTFilterBank *bank) { return bank ? bank->state_ld : 0; }

int RTFilterBank_set_mode(RTFilterBank *bank, int mode)
{
    if (!bank) return fail(RTFB_EINVAL, "NULL bank");
    if (mode != RTFB_MODE_AUTO && mode != RTFB_MODE_SEQUENTIAL && mode != RTFB_MODE_SCAN)
        return fail(RTFB_EINVAL, "mode is not an RTFB_MODE_* value");
    bank->mode = mode;
    return RTFB_OK;
}

/* Does a call of n samples per row on this one-group bank go to iir_scan16_kernel on whole aligned segments?  (The
 * dispatch below, restated for RTFilterStream, which may then leave the run-ups of its sub-chunks to that kernel.) */
static bool scan16_serves(const RTFilterBank *bank, size_t n)
{
    if (bank->groups.size() != 1) return false;
    const Group &g = bank->groups[0];
    if (g.cls != kIirFast || g.sos_sections || g.order <= kScan2MaxOrder || !iir_scan_supported(g.order)) return false;
    if (bank->mode == RTFB_MODE_SEQUENTIAL || !g.has_tables || (bank->mode == RTFB_MODE_AUTO && !g.scan_safe)) return false;
    return !pipe16_enabled() && n >= (size_t)kWave * kScanT && n % ((size_t)kWave * kScanT) == 0;
}

static int bank_updaten(RTFilterBank *bank, double *out, size_t ld_out, const double *in,
                        size_t ld_in, size_t n, void *hip_stream, bool no_alloc)
{
    if (!bank || !out || !in) return fail(RTFB_EINVAL, "RTFilterBank_updaten: NULL argument");
    if (n == 0) return RTFB_OK;
    if (bank->nch > 1 && (ld_in < n || ld_out < n))
        return fail(RTFB_EINVAL, "RTFilterBank_updaten: ld smaller than n");
    DeviceScope ds(bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);

    BlockIO io;
    io.out = out;
    io.ld_out = ld_out;
    io.in = in;
    io.ld_in = ld_in;
    io.n = n;
    io.aligned16 = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16 == 0) &&
                   ld_in % 2 == 0 && ld_out % 2 == 0;

    {
        int rc = chain_close(bank, s);
        if (rc) return rc;
    }
    if (bank->stream_rows && !(scan16_serves(bank, n) && io.aligned16))       /* before anything is enqueued */
        return fail(RTFB_EINVAL, "internal: stream rows handed to a call the order 9..16 scan kernel does not serve whole");
    const double *in_end = in + (bank->nch - 1) * ld_in + n;
    const double *out_end = out + (bank->nch - 1) * ld_out + n;
    const bool overlap = in < out_end && out < in_end;
    if (no_alloc && overlap && bank->nch * n > kStaticScratchDoubles) {
        /* refused BEFORE anything is enqueued: groups launched ahead of the FIR group would already have
         * advanced their channels */
        for (const Group &g : bank->groups)
            if (g.cls == kFir)
                return fail(RTFB_ESIZE, "RTFilterBank_updaten_noalloc: an in-place call on FIR channels stages "
                                        "channels x n doubles; the static scratch holds " +
                                            std::to_string(kStaticScratchDoubles));
    }
    const BlockIO io_first = io;
    for (Group &g : bank->groups) {
        GroupView v = view_of(bank, g);
        io = io_first;
        if (g.stage > 0) {                   /* a later section of SOS channels filters the output rows in place */
            io.in = out;
            io.ld_in = ld_out;
            io.aligned16 = (reinterpret_cast<uintptr_t>(out) % 16 == 0) && ld_out % 2 == 0;
        }
        if (g.cls == kFir) {
            /* FIR outputs are computed out of order in time, so the kernels must not see their
             * own outputs as inputs: an in-place call (allowed by the reference, where each
             * sample is read before its slot is written) reads from a private copy instead. */
            BlockIO fio = io;
            if (overlap && no_alloc) {
                /* staging through this GPU's static array: no pool, no hipMalloc (size checked above) */
                HIP_TRY(hipMemcpy2DAsync(bank->d_static_scratch, n * sizeof(double), in, ld_in * sizeof(double),
                                         n * sizeof(double), bank->nch, hipMemcpyDeviceToDevice, s));
                fio.in = bank->d_static_scratch;
                fio.ld_in = n;
            } else if (overlap) {
                const size_t need = bank->nch * n;
                if (g.scratch_doubles < need) {
                    HIP_TRY(hipStreamSynchronize(s));
                    pool_free(g.d_scratch);
                    g.d_scratch = nullptr;
                    g.scratch_doubles = 0;
                    void *p = nullptr;
                    int rc = pool_alloc(&p, need * sizeof(double), bank->device);
                    if (rc) return rc;
                    g.d_scratch = static_cast<double *>(p);
                    g.scratch_doubles = need;
                }
                HIP_TRY(hipMemcpy2DAsync(g.d_scratch, n * sizeof(double), in, ld_in * sizeof(double),
                                         n * sizeof(double), bank->nch, hipMemcpyDeviceToDevice, s));
                fio.in = g.d_scratch;
                fio.ld_in = n;
            }
            v.fir_mfma = (bank->mode != RTFB_MODE_SEQUENTIAL && g.kmax >= kFirMfmaMinTaps) ? 1 : 0;
            HIP_TRY(launch_fir(v, fio, s));      /* leaves the end state in g.d_state (d_state_alt is scratch) */
            continue;
        }
        if (g.cls == kIirGeneric) {
            HIP_TRY(launch_iir_generic(v, io, s));
            continue;
        }
        /* AUTO: from 65 536 channels the bit-exact lane-per-channel kernel fills the GPU and streams as
         * fast as the scan kernels (tools/seq_sweep.py, order 8: 4.3 vs 4.7 TB/s at 65 536 x 16K, 4.9 vs
         * 3.9 TB/s at 262 144 x 4K; order 16: 3.6 vs 2.8 TB/s; short blocks: 4.5 vs 0.3 TB/s at 2^20 x 256);
         * below that (3.3 vs 4.7 TB/s at 32 768 channels) the time-parallel kernels win.  Orders <= 8 switch
         * one octave later: at exactly 65 536 channels x 64K the pipelined kernel still leads (4.6 vs 3.7 TB/s).
         * Orders 9..16: the matrix-core scan kernel leads at every width once a call holds a full 4096-sample
         * segment (4.6 vs 3.7 TB/s at 65 536 x 16K, 4.1 vs 3.7 at 262 144 x 4K); shorter calls of wide banks
         * stay on the lane-per-channel kernel. */
        const size_t seq_from = g.order <= kScan2MaxOrder ? 2 * kAutoSeqMinChannels
                                : (n >= (size_t)kWave * kScanT ? (size_t)-1 : kAutoSeqMinChannels);
        bool seq = bank->mode == RTFB_MODE_SEQUENTIAL || !g.has_tables ||
                   (bank->mode == RTFB_MODE_AUTO && (!g.scan_safe || g.chans.size() >= seq_from));
        if (!seq && !iir_scan_supported(g.order)) seq = true;
        if (g.sos_sections) {
            /* fused cascades: all sections per sample, one pass (bit-exact lane-per-channel kernel or the
             * pipelined kernel on the cascade's 2-states-per-section system) */
            if (seq) HIP_TRY(launch_iir_seq_sos(g.sos_sections, v, io, s));
            else HIP_TRY(launch_iir_pipe_sos(g.sos_sections, v, io, s));
            continue;
        }
        if (seq) HIP_TRY(launch_iir_seq(g.order, v, io, s));
        else if (g.order <= kScan2MaxOrder && bank->designs.size() <= kMaxPipeDesigns)
            HIP_TRY(launch_iir_pipe(g.order, v, io, s));            /* per-design row propagators resident */
        else if (g.order <= kScan2MaxOrder) HIP_TRY(launch_iir_scan2(g.order, v, io, s));   /* > 1024 designs */
        else if (pipe16_enabled() && io.aligned16 && n % kPipe16Seg == 0)
            HIP_TRY(launch_iir_pipe16(g.order, v, io, Pipe16Ext{}, 0, s));   /* orders 9..16, whole segments */
        else {
            io.runup = bank->stream_rows;                           /* (checked above) */
            io.runup_src = bank->stream_rows ? bank->stream_runup_src : nullptr;
            HIP_TRY(launch_iir_scan(g.order, v, io, s));            /* orders 9..16, any shape */
        }
    }
    return RTFB_OK;
}

int RTFilterBank_updaten(RTFilterBank *bank, double *out, size_t ld_out, const double *in,
                         size_t ld_in, size_t n, void *hip_stream)
{
    return bank_updaten(bank, out, ld_out, in, ld_in, n, hip_stream, false);
}

int RTFilterBank_updaten_noalloc(RTFilterBank *bank, double *out, size_t ld_out, const double *in,
                                 size_t ld_in, size_t n, void *hip_stream)
{
    return bank_updaten(bank, out, ld_out, in, ld_in, n, hip_stream, true);
}

size_t RTFilterBank_noalloc_max_doubles(void) { return kStaticScratchDoubles; }

/* Period of a paced tiled launch in 1/1024 ticks of 10 ns, 0 = none.  Why: in the tiled layout the workgroups that
 * run together sweep one contiguous window of HBM per step -- as long as they ARE in step.  Left alone they drift
 * apart (workgroups on some XCDs run 3-5 % faster) and the lock-step rate of the memory system is lost: the copy
 * twin falls from 5.9 to 5.5 TB/s when its workgroups are up to 4 segments out of step, to 5.0 without any order
 * (profiles/r3_tiled_copy.txt); a launch of 256K samples per channel runs 12 % faster on a schedule than without,
 * one of 64K 4-6 % (profiles/r3_pacing.txt).  The schedule: step `it` not before launch start + it x period, the
 * period being what one round of segments (32 KiB per workgroup, read + written) takes at the lock-step copy rate.
 * A period a few per cent too short loses the effect on long launches (every workgroup falls behind and runs
 * free), one too long costs its excess; the constant below is the rate of a fast-state MI355X of this pool.
 * Only for groups that fill at least 3/4 of one resident round and launches of 12 segments or more: narrower
 * banks are not bandwidth-bound (no workgroup is ever early, and the clock reads cost the front wave 0.3 us per
 * step), shorter launches are mostly pipeline fill. */
constexpr double kPaceCopyGBs = 5950.0;      /* lock-step 1:1 copy of the tiled layout on MI355X */

unsigned int pace_period(const Group &g, int count, size_t n, int round)
{
    if (g.pace_ns == 0.0 || count > round || 4 * count < 3 * round || n / RTFB_TILE < 12) return 0;
    const double ns = g.pace_ns > 0.0 ? g.pace_ns : (double)count * 32768.0 / kPaceCopyGBs;
    return (unsigned int)(ns / 10.0 * 1024.0);
}

/* Shared by _updaten_tiled and _updaten_chained: argument checks, the block's addressing.  tiled: in / out are
 * [n / 2048][nch][2048] (ld ignored). */
static int pipe_block(const RTFilterBank *bank, double *out, size_t ld_out, const double *in, size_t ld_in, size_t n,
                      bool tiled, BlockIO *io, PipeExt *ext)
{
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16) != 0)
        return fail(RTFB_EINVAL, "in / out must be 16-byte aligned");
    if (n % RTFB_TILE) return fail(RTFB_ESIZE, "n must be a multiple of RTFB_TILE (2048) samples");
    io->out = out;
    io->in = in;
    io->n = n;
    io->aligned16 = 1;
    *ext = PipeExt();
    if (tiled) {
        io->ld_in = io->ld_out = RTFB_TILE;
        ext->seg_in = ext->seg_out = bank->nch * (size_t)RTFB_TILE;
    } else {
        if (bank->nch > 1 && (ld_in < n || ld_out < n)) return fail(RTFB_EINVAL, "ld smaller than n");
        if (ld_in % 2 || ld_out % 2) return fail(RTFB_EINVAL, "row pitches must be even (16-byte aligned rows)");
        io->ld_in = ld_in;
        io->ld_out = ld_out;
        ext->seg_in = ext->seg_out = RTFB_TILE;
    }
    return RTFB_OK;
}

/* FIR banks in the tiled block layout: ONE launch group that the matrix-core kernel serves with every head folded
 * (stable_init designs of 2..2047 taps, not RTFB_MODE_SEQUENTIAL), n a multiple of 4096.  Same results as
 * RTFilterBank_updaten on the same blocks (same kernel arithmetic); not paced (fir_kernels.hip: why). */

static int fir_updaten_tiled(RTFilterBank *bank, double *out, const double *in, size_t n, void *hip_stream)
{
    const char *what = "the tiled block layout is served by the pipelined IIR kernel (one launch group of order <= 8, "
                       "<= 1024 designs, scan-safe or RTFB_MODE_SCAN) and by the FIR kernels (one launch group, n a "
                       "multiple of 4096: 2..32 taps in every mode, stable_init designs of 33..2047 taps outside "
                       "RTFB_MODE_SEQUENTIAL)";
    if (bank->groups.size() != 1 || bank->groups[0].cls != kFir) return fail(RTFB_EUNSUPPORTED, what);
    const bool exact = bank->mode == RTFB_MODE_SEQUENTIAL;
    Group &g = bank->groups[0];
    DeviceScope ds(bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16) != 0)
        return fail(RTFB_EINVAL, "in / out must be 16-byte aligned");
    GroupView v = view_of(bank, g);
    v.fir_mfma = g.kmax >= kFirMfmaMinTaps ? 1 : 0;
    if (!fir_tiled_supported(v, n, exact)) return fail(n % 4096 ? RTFB_ESIZE : RTFB_EUNSUPPORTED, what);
    int rc = chain_close(bank, s);
    if (rc) return rc;
    BlockIO io;
    io.out = out;
    io.in = in;
    io.ld_in = io.ld_out = RTFB_TILE;
    io.n = n;
    io.aligned16 = 1;
    const size_t seg = bank->nch * (size_t)RTFB_TILE, total = bank->nch * n;
    if (in == out) {
        /* FIR outputs are produced out of time order: an in-place call reads from a private copy (same layout) */
        if (g.scratch_doubles < total) {
            HIP_TRY(hipStreamSynchronize(s));
            pool_free(g.d_scratch);
            g.d_scratch = nullptr;
            g.scratch_doubles = 0;
            void *p = nullptr;
            rc = pool_alloc(&p, total * sizeof(double), bank->device);
            if (rc) return rc;
            g.d_scratch = static_cast<double *>(p);
            g.scratch_doubles = total;
        }
        HIP_TRY(hipMemcpyAsync(g.d_scratch, in, total * sizeof(double), hipMemcpyDeviceToDevice, s));
        io.in = g.d_scratch;
    } else if (in < out + total && out < in + total)
        return fail(RTFB_EINVAL, "RTFilterBank_updaten_tiled: in and out overlap without being equal");
    HIP_TRY(launch_fir_tiled(v, io, seg, seg, exact, g.pace_ns, s));
    return RTFB_OK;
}

int RTFilterBank_updaten_tiled(RTFilterBank *bank, double *out, const double *in, size_t n, void *hip_stream)
{
    if (!bank || !out || !in) return fail(RTFB_EINVAL, "RTFilterBank_updaten_tiled: NULL argument");
    if (n == 0) return RTFB_OK;
    Group *g = chain_group(bank);
    if (!g) return fir_updaten_tiled(bank, out, in, n, hip_stream);
    DeviceScope ds(bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    BlockIO io;
    PipeExt ext;
    int rc = pipe_block(bank, out, 0, in, 0, n, true, &io, &ext);
    if (rc || (rc = chain_close(bank, s))) return rc;
    const GroupView v = view_of(bank, *g);
    const int round = pipe_resident_round();
    /* A group wider than one resident round goes round by round: in this layout the workgroups of ONE round sweep
     * one contiguous window per step, a launch several rounds deep would put workgroups of different segments (a
     * whole [nch][2048] slab apart) next to each other. */
    const int step = v.count > round ? round : v.count;
    for (int off = 0; off < v.count; off += step) {
        GroupView h = v;
        h.chan = v.chan ? v.chan + off : nullptr;
        h.chan0 = v.chan0 + off;
        h.count = std::min(step, v.count - off);
        ext.pace_fx = pace_period(*g, h.count, n, round);
        h.design = v.design + off;
        h.state = v.state + (size_t)off * v.state_ld;
        h.fval = v.fval + off;
        h.init = v.init + off;
        HIP_TRY(launch_iir_pipe_ext(g->order, h, io, ext, s));
    }
    return RTFB_OK;
}

int RTFilterBank_updaten_chained(RTFilterBank *bank, double *out, size_t ld_out, const double *in, size_t ld_in,
                                 size_t n, unsigned flags, void *hip_stream, void **done_event)
{
    if (!bank || !out || !in) return fail(RTFB_EINVAL, "RTFilterBank_updaten_chained: NULL argument");
    if (flags & ~(unsigned)RTFB_CHAIN_TILED) return fail(RTFB_EINVAL, "RTFilterBank_updaten_chained: unknown flag");
    if (done_event) *done_event = nullptr;
    if (n == 0) return RTFB_OK;
    const bool tiled = (flags & RTFB_CHAIN_TILED) != 0;
    DeviceScope ds(bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    Group *g = chain_group(bank);
    const bool chainable = g && (int)g->chans.size() <= pipe_resident_round() && n % RTFB_TILE == 0 &&
                           ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16) == 0 &&
                           (tiled || (ld_in % 2 == 0 && ld_out % 2 == 0));
    if (!chainable) {
        /* stream order: the same results, no overlap with the call before */
        int rc = tiled ? RTFilterBank_updaten_tiled(bank, out, in, n, hip_stream)
                       : RTFilterBank_updaten(bank, out, ld_out, in, ld_in, n, hip_stream);
        if (rc || !done_event) return rc;
        if ((rc = chain_prepare_events(bank))) return rc;
        ChainState &c = bank->chain;
        hipEvent_t ev = c.done[c.seq % kChainRing];
        c.seq++;
        HIP_TRY(hipEventRecord(ev, s));
        *done_event = ev;
        return RTFB_OK;
    }
    BlockIO io;
    PipeExt ext;
    int rc = pipe_block(bank, out, ld_out, in, ld_in, n, tiled, &io, &ext);
    if (rc || (rc = chain_prepare(bank, *g))) return rc;
    ChainState &c = bank->chain;
    if (*c.h_error) return chain_close(bank, s);
    const int slot = (int)(c.seq % kChainRing), prev = (int)((c.seq + kChainRing - 1) % kChainRing);
    hipStream_t x = c.s[c.seq & 1];
    /* the link is ordered behind everything enqueued on the caller's stream so far (its input block) */
    HIP_TRY(hipEventRecord(c.in_ready[slot], s));
    HIP_TRY(hipStreamWaitEvent(x, c.in_ready[slot], 0));
    ext.chain.my_queue = c.d_queue + (size_t)slot * c.count;
    ext.chain.finished = c.d_counters + 16 * slot;
    ext.chain.finished_base = c.finished_total[slot];
    ext.chain.started = c.d_counters + 16 * (kChainRing + slot);
    ext.chain.records = c.d_records;
    ext.chain.gen = (unsigned int)(c.seq + 1);
    ext.chain.error = c.d_error;
    if (c.open) {
        /* Gate: this launch may only be dispatched once every workgroup of the link before it (other stream) is
         * resident -- and that link is itself behind its own input event, which this stream therefore waits for
         * too, so that the gate never spins on a producer of the caller's. */
        HIP_TRY(hipStreamWaitEvent(x, c.in_ready[prev], 0));
        HIP_TRY(launch_chain_gate(c.d_counters + 16 * (kChainRing + prev), c.started_total[prev], c.d_error, x));
        ext.chain.prev_queue = c.d_queue + (size_t)prev * c.count;
    } else {
        ext.chain.prev_queue = nullptr;     /* head of a chain: ordered behind `s` by the event above */
    }
    c.started_total[slot] += (unsigned long long)c.count;
    c.finished_total[slot] += (unsigned long long)c.count;
    HIP_TRY(launch_iir_pipe_ext(g->order, view_of(bank, *g), io, ext, x));
    HIP_TRY(hipEventRecord(c.done[slot], x));
    c.open = true;
    c.links++;
    c.seq++;
    if (done_event) *done_event = c.done[slot];
    return RTFB_OK;
}

int RTFilterBank_chain_join(RTFilterBank *bank, void *hip_stream)
{
    if (!bank) return fail(RTFB_EINVAL, "NULL bank");
    return chain_close(bank, static_cast<hipStream_t>(hip_stream));
}

int RTFilterBank_set_pace(RTFilterBank *bank, double ns_per_segment)
{
    if (!bank) return fail(RTFB_EINVAL, "NULL bank");
    if (ns_per_segment > 1e6) return fail(RTFB_EINVAL, "RTFilterBank_set_pace: period above 1 ms");
    for (Group &g : bank->groups) g.pace_ns = ns_per_segment;
    return RTFB_OK;
}

int dsp_tile_rows(double *tiles, const double *rows, size_t ld, size_t nch, size_t n, void *hip_stream)
{
    if (!tiles || !rows) return fail(RTFB_EINVAL, "dsp_tile_rows: NULL argument");
    if (n % RTFB_TILE || ld < n) return fail(RTFB_EINVAL, "dsp_tile_rows: n a multiple of RTFB_TILE, ld >= n");
    HIP_TRY(launch_tile_rows(tiles, rows, ld, nch, n, 1, static_cast<hipStream_t>(hip_stream)));
    return RTFB_OK;
}

int dsp_untile_rows(double *rows, size_t ld, const double *tiles, size_t nch, size_t n, void *hip_stream)
{
    if (!tiles || !rows) return fail(RTFB_EINVAL, "dsp_untile_rows: NULL argument");
    if (n % RTFB_TILE || ld < n) return fail(RTFB_EINVAL, "dsp_untile_rows: n a multiple of RTFB_TILE, ld >= n");
    HIP_TRY(launch_tile_rows(rows, tiles, ld, nch, n, 0, static_cast<hipStream_t>(hip_stream)));
    return RTFB_OK;
}

/* RTFilterBank_updaten on HOST arrays: for c: RTFilter_updaten(out + c*ld_out, filter_c, in + c*ld_in, n) as a
 * caller of the reference makes it, moved through the GPU in chunks of time (all channels x nt samples, so that
 * the carried state streams), through pinned staging rings.  Bounded by the host link and the CPU's copies. */
int RTFilterBank_updaten_host(RTFilterBank *bank, double *out, size_t ld_out, const double *in, size_t ld_in, size_t n)
{
    if (!bank || !out || !in) return fail(RTFB_EINVAL, "RTFilterBank_updaten_host: NULL argument");
    if (n == 0) return RTFB_OK;
    if (bank->nch > 1 && (ld_in < n || ld_out < n))
        return fail(RTFB_EINVAL, "RTFilterBank_updaten_host: ld smaller than n");
    DeviceScope ds(bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    /* chunk: all channels x nt samples in one staging buffer; whole 2048-sample segments when they fit */
    size_t nt = kHostChunkBytes / sizeof(double) / bank->nch;
    if (nt >= 2048) nt = nt / 2048 * 2048;
    if (nt == 0) nt = 1;
    nt = std::min(nt, n);
    int rc = host_pipe_prepare(bank->host, bank->device, std::max(bank->nch * nt * sizeof(double), (size_t)4096));
    if (rc) return rc;
    return host_pipe_run(bank->host, bank->nch, n, in, ld_in, out, ld_out, nt,
                         [&](double *d_out, const double *d_in, size_t len, hipStream_t s) {
                             return RTFilterBank_updaten(bank, d_out, len, d_in, len, len, s);
                         });
}

int dsp_copy_rows(double *out, size_t ld_out, const double *in, size_t ld_in, size_t nch, size_t n,
                  int threads, int seg, void *hip_stream)
{
    if (!out || !in) return fail(RTFB_EINVAL, "dsp_copy_rows: NULL argument");
    if ((threads != 64 && threads != 128 && threads != 256) || seg <= 0 || seg % (2 * threads) ||
        seg / 2 / threads > 16 || n % (size_t)seg)
        return fail(RTFB_EINVAL, "dsp_copy_rows: threads in {64,128,256}, seg a multiple of 2*threads, "
                                 "at most 32*threads, dividing n");
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16) || ld_in % 2 || ld_out % 2)
        return fail(RTFB_EINVAL, "dsp_copy_rows: needs 16-byte aligned rows");
    HIP_TRY(launch_copy_rows(out, ld_out, in, ld_in, nch, n, threads, seg, 0, static_cast<hipStream_t>(hip_stream)));
    return RTFB_OK;
}

/* the same twin for the tiled block layout [n / RTFB_TILE][nch][RTFB_TILE]: workgroup c walks tile (s, c), s ascending */
int dsp_copy_tiles(double *out, const double *in, size_t nch, size_t n, void *hip_stream)
{
    if (!out || !in) return fail(RTFB_EINVAL, "dsp_copy_tiles: NULL argument");
    if (n % RTFB_TILE || ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16))
        return fail(RTFB_EINVAL, "dsp_copy_tiles: n a multiple of RTFB_TILE, 16-byte aligned blocks");
    HIP_TRY(launch_copy_rows(out, RTFB_TILE, in, RTFB_TILE, nch, n, 256, RTFB_TILE, nch * (size_t)RTFB_TILE,
                             static_cast<hipStream_t>(hip_stream)));
    return RTFB_OK;
}

/* Diagnostics (DSP_AMD_SCAN_STAMPS=1): copy the scan kernel's per-channel phase-cycle sums
 * (32 uint64 per channel of the first fast-IIR group; scan2 uses a stride of 8) to the host. Not declared in the headers. */
EXPORT int dsp_debug_scan_stamps(RTFilterBank *bank, uint64_t *dst, size_t nwords)
{
    if (!bank || !dst) return fail(RTFB_EINVAL, "NULL argument");
    DeviceScope ds(bank->device);
    HIP_TRY(hipDeviceSynchronize());
    for (const Group &g : bank->groups)
        if (g.cls == kIirFast && g.d_aux) {
            const size_t have = g.chans.size() * 32;
            HIP_TRY(hipMemcpy(dst, g.d_aux, std::min(have, nwords) * 8, hipMemcpyDeviceToHost));
            return RTFB_OK;
        }
    return fail(RTFB_EINVAL, "no stamped group (set DSP_AMD_SCAN_STAMPS=1 before creating the bank)");
}

int RTFilterBank_reset(RTFilterBank *bank, void *hip_stream)
{
    if (!bank) return fail(RTFB_EINVAL, "NULL bank");
    DeviceScope ds(bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    {
        int rc = chain_close(bank, static_cast<hipStream_t>(hip_stream));
        if (rc) return rc;
    }
    for (const Group &g : bank->groups)
        HIP_TRY(launch_reset(view_of(bank, g), static_cast<hipStream_t>(hip_stream)));
    return RTFB_OK;
}

int RTFilterBank_get_state(RTFilterBank *bank, double *state, double *fval, int32_t *init,
                           void *hip_stream)
{
    if (!bank) return fail(RTFB_EINVAL, "NULL bank");
    DeviceScope ds(bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    {
        int rc = chain_close(bank, static_cast<hipStream_t>(hip_stream));
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)));
    if (bank->chain.h_error && *bank->chain.h_error) return chain_close(bank, static_cast<hipStream_t>(hip_stream));
    const size_t ld = bank->state_ld;
    for (const Group &g : bank->groups) {
        const size_t cnt = g.chans.size();
        std::vector<double> st(cnt * (size_t)g.state_ld), fv(cnt);
        std::vector<int32_t> ini(cnt);
        HIP_TRY(hipMemcpy(st.data(), g.d_state, st.size() * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(fv.data(), g.d_fval, cnt * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ini.data(), g.d_init, cnt * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < cnt; ++i) {
            const size_t c = (size_t)g.chans[i];
            const Design &cd = bank->designs[bank->design_of[c]];
            if (cd.kind == RTFB_KIND_SOS && !cd.sos_fused) {
                /* section `stage` owns slots [3 stage, 3 stage + 3); the last section's output is the channel's
                 * filtered_value; every section carries the same `initialized` */
                const size_t off = (size_t)3 * g.stage;
                const bool last = g.stage + 1 == (int)cd.section_design.size();
                if (state) {
                    for (size_t k = 0; k < 3; ++k) state[c * ld + off + k] = st[i * g.state_ld + k];
                    if (last)
                        for (size_t k = off + 3; k < ld; ++k) state[c * ld + k] = 0.0;
                }
                if (fval && last) fval[c] = fv[i];
                if (init && g.stage == 0) init[c] = ini[i];
                continue;
            }
            if (state) {
                for (size_t k = 0; k < ld; ++k)
                    state[c * ld + k] = k < (size_t)g.state_ld ? st[i * g.state_ld + k] : 0.0;
            }
            if (fval) fval[c] = fv[i];
            if (init) init[c] = ini[i];
        }
    }
    return RTFB_OK;
}

int RTFilterBank_set_state(RTFilterBank *bank, const double *state, const double *fval,
                           const int32_t *init, void *hip_stream)
{
    if (!bank) return fail(RTFB_EINVAL, "NULL bank");
    DeviceScope ds(bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    {
        int rc = chain_close(bank, static_cast<hipStream_t>(hip_stream));
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)));
    const size_t ld = bank->state_ld;
    for (const Group &g : bank->groups) {
        const size_t cnt = g.chans.size();
        if (state) {
            std::vector<double> st(cnt * (size_t)g.state_ld);
            for (size_t i = 0; i < cnt; ++i) {
                const size_t c = (size_t)g.chans[i];
                const Design &cd = bank->designs[bank->design_of[c]];
                const size_t off = (cd.kind == RTFB_KIND_SOS && !cd.sos_fused) ? (size_t)3 * g.stage : 0;
                for (int k = 0; k < g.state_ld; ++k)
                    st[i * g.state_ld + k] = off + k < ld ? state[c * ld + off + k] : 0.0;
            }
            HIP_TRY(hipMemcpy(g.d_state, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        if (fval) {
            std::vector<double> fv(cnt);
            for (size_t i = 0; i < cnt; ++i) fv[i] = fval[g.chans[i]];
            HIP_TRY(hipMemcpy(g.d_fval, fv.data(), cnt * sizeof(double), hipMemcpyHostToDevice));
        }
        if (init) {
            std::vector<int32_t> ini(cnt);
            for (size_t i = 0; i < cnt; ++i) ini[i] = init[g.chans[i]];
            HIP_TRY(hipMemcpy(g.d_init, ini.data(), cnt * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    return RTFB_OK;
}

int RTFilterBank_sync_to_host(RTFilterBank *bank, void *hip_stream)
{
    if (!bank) return fail(RTFB_EINVAL, "NULL bank");
    if (bank->host_filters.empty())
        return fail(RTFB_EINVAL, "bank was not built from host filters");
    const size_t ld = bank->state_ld;
    std::vector<double> st(bank->nch * ld), fv(bank->nch);
    std::vector<int32_t> ini(bank->nch);
    int rc = RTFilterBank_get_state(bank, st.data(), fv.data(), ini.data(), hip_stream);
    if (rc) return rc;
    for (size_t c = 0; c < bank->nch; ++c) {
        RTFilter *f = bank->host_filters[c];
        const Design &d = bank->designs[bank->design_of[c]];
        double *dst = d.kind == RTFB_KIND_FIR || d.kind == RTFB_KIND_FIR_PARTIAL
                          ? reinterpret_cast<RTFIRFilter *>(f)->state
                          : reinterpret_cast<RTIIRFilter *>(f)->state;
        if (d.kind == RTFB_KIND_SOS) {              /* the host filter keeps two doubles per section */
            for (size_t k = 0; k < (size_t)d.nb / 3; ++k) {
                dst[2 * k] = st[c * ld + 3 * k];
                dst[2 * k + 1] = st[c * ld + 3 * k + 1];
            }
        } else
            std::memcpy(dst, &st[c * ld], d.state_slots() * sizeof(double));
        f->filtered_value = fv[c];
        f->initialized = ini[c];
    }
    return RTFB_OK;
}

}  /* extern "C" */

/* ======================================================================== */
/* RTFilterStream: one filter over a very long stream, time-parallel          */
/* (include/RTFilterStream.h; BASELINE.json config 5)                         */

namespace {

constexpr size_t kStreamMaxSplit = 1024;      /* sub-chunks per call at most while they are long: the 1024 workgroups (4
                                                 per CU) the GPU holds at once; with 2048 the second half only starts as
                                                 the first drains: -1.2 % */
/* Long streams: MORE, SHORTER sub-chunks.  1024 sub-chunks of a 2^30-sample stream are 1024 windows 8 MiB apart, each
 * on its own pages, read and written: a plain copy with that pattern reaches 4.85-5.17 TB/s (tools/
 * stream16_copy_ceiling.py), and the order-16 kernel runs at exactly that, 4.94.  Sub-chunks of 2^17 samples, taken
 * by the workgroups in dispatch order, 1024 at a time, make the resident workgroups sweep ONE window of 1 GiB that
 * slides along the stream: 5.27 TB/s (2^18: 5.20, 2^16: 5.23, 2^14: 4.95 -- a workgroup's first segment is not
 * overlapped; tools/stream16_split_shape.py; config 5 end to end by sub-chunk length: 2^15 310 000 Msamples/s, 2^16
 * 319 500, 2^17 326 000, 2^18 322 400, 2^19 318 300; one resident round of waves walking the sub-chunks with the
 * prefetch running across them instead of a wave per sub-chunk: the same 326 600 at 2^17, less at every other
 * length -- not kept).  A run-up LAUNCH over 8191 sub-chunks (one short ragged segment per
 * workgroup, eight rounds of them) costs more than that gains (config 5: 310 000 -> 292 000 Msamples/s), so the
 * scan kernel of orders 9..16 computes the run-ups itself (BlockIO::runup, kernels.hip): 332 000. */
constexpr size_t kStreamChunkTarget = (size_t)1 << 17;
constexpr size_t kStreamMaxSplitShort = 65536;
constexpr size_t kScanSeg = (size_t)kWave * kScanT;           /* iir_scan16_kernel's segment: its run-up length */
constexpr size_t kStreamMinSplit = 8;         /* below this the sequential path is as good */
constexpr size_t kStreamGrain = 2048;         /* sub-chunk lengths are multiples of the scan segment */
constexpr size_t kStreamMaxDecay = (size_t)1 << 20;
constexpr long double kStreamNegligible = 0x1.0p-80L;
constexpr size_t kStreamExactChunk = 16384;   /* exact composition: sub-chunks of at least this many samples */

struct SplitBanks {
    RTFilterBank *main = nullptr;             /* C channels: the sub-chunks */
    RTFilterBank *warm = nullptr;             /* C-1 channels: zero-state run-ups to sub-chunks 1..C-1 */
};

}  // namespace

struct RTFilterStream {
    int device = 0;
    std::vector<double> b, a;
    int kind = RTFB_KIND_IIR;
    size_t decay = 0;                         /* M; 0 = no time split */
    size_t last_split = 1;
    RTFilterBank *one = nullptr;              /* 1 channel: owns the carried state; sequential path */
    RTFilterBank *tail = nullptr;             /* 1 channel: RTFilterStream_tail_state */
    std::map<size_t, SplitBanks> split;       /* by C */
    double *d_scratch = nullptr;              /* outputs of the run-ups (discarded) */
    size_t scratch_doubles = 0;
    bool exact_ok = false;                    /* chunk maps may be composed exactly (scan-safe fast IIR) */
    bool maybe_fresh = true;                  /* the next sample may go to the initialiser (host-side view) */
    std::map<size_t, double *> d_power;       /* chunk length L -> A^L (N x N, row-major) on the device */
    /* interleaved segments (orders 9..16, pipe16_kernel.hip stream mode) */
    double *d_exch = nullptr;                 /* granules {value, generation}: 16 per segment of a call */
    size_t exch_segments = 0;
    unsigned long long exch_gen = 0;          /* generations handed out so far: never reused on this buffer */
    unsigned int *h_error = nullptr;          /* pinned, device-visible: a bounded wait of the kernel ran out */
    int interleave = 0;                       /* workgroups to deal the segments to; <= 0 = mode off (default) */
    HostPipe host;                            /* staging rings of RTFilterStream_updaten_host, built on first use */
    /* RTFilterStream_split_updaten: this rank's two hand-off vectors and everybody's, kept with the stream */
    double *d_split_mine = nullptr, *d_split_all = nullptr;
    size_t split_world = 0;
    bool split_carry_remote = false;          /* the stream's end state lives on the last rank of the last split call */
};

namespace {

RTFilterDesign design_view(const RTFilterStream *st)
{
    RTFilterDesign d;
    d.kind = st->kind;
    d.nb = st->b.size();
    d.na = st->a.size();
    d.b = st->b.data();
    d.a = st->a.empty() ? nullptr : st->a.data();
    return d;
}

/* Smallest power of two M with max|A^M| < 2^-80 (long double, by squaring); 0 if none up to 2^20. */
size_t decay_length(const std::vector<double> &a)
{
    const int N = (int)a.size() - 1;
    if (N < 1) return 0;
    typedef std::vector<long double> Mat;
    Mat P((size_t)N * N, 0.0L);
    for (int i = 0; i < N; ++i) {
        P[(size_t)i * N] = -(long double)a[i + 1] / (long double)a[0];
        if (i + 1 < N) P[(size_t)i * N + i + 1] += 1.0L;
    }
    for (size_t m = 1; m <= kStreamMaxDecay; m <<= 1) {
        long double big = 0.0L;
        for (long double v : P) big = std::max(big, std::fabs(v));
        if (!(big == big) || big > 0x1.0p+400L) return 0;       /* unstable */
        if (big < kStreamNegligible) return std::max<size_t>(m, 64);
        Mat Q((size_t)N * N, 0.0L);
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < N; ++k) {
                const long double pik = P[(size_t)i * N + k];
                for (int j = 0; j < N; ++j) Q[(size_t)i * N + j] += pik * P[(size_t)k * N + j];
            }
        P.swap(Q);
    }
    return 0;
}

int one_channel_bank(RTFilterBank **bank, const RTFilterStream *st)
{
    const RTFilterDesign d = design_view(st);
    const int32_t zero = 0;
    return RTFilterBank_from_designs(bank, &d, 1, &zero, 1, st->device);
}

int split_banks(RTFilterStream *st, size_t C, SplitBanks **out)
{
    auto it = st->split.find(C);
    if (it == st->split.end()) {
        SplitBanks sb;
        const RTFilterDesign d = design_view(st);
        std::vector<int32_t> zeros(C, 0);
        int rc = RTFilterBank_from_designs(&sb.main, &d, 1, zeros.data(), C, st->device);
        if (rc) return rc;
        rc = RTFilterBank_from_designs(&sb.warm, &d, 1, zeros.data(), C - 1, st->device);
        if (rc) {
            RTFilterBank_destroy(sb.main);
            return rc;
        }
        it = st->split.emplace(C, sb).first;
    }
    *out = &it->second;
    return RTFB_OK;
}

int stream_scratch(RTFilterStream *st, size_t doubles, hipStream_t s)
{
    if (st->scratch_doubles >= doubles) return RTFB_OK;
    HIP_TRY(hipStreamSynchronize(s));
    pool_free(st->d_scratch);
    st->d_scratch = nullptr;
    st->scratch_doubles = 0;
    void *p = nullptr;
    int rc = pool_alloc(&p, doubles * sizeof(double), st->device);
    if (rc) return rc;
    st->d_scratch = static_cast<double *>(p);
    st->scratch_doubles = doubles;
    return RTFB_OK;
}

/* One time-split piece: in[0 .. C*Lc) as C channels of Lc samples. */
int stream_split_piece(RTFilterStream *st, double *out, const double *in, size_t C, size_t Lc, bool self_runup,
                       hipStream_t s)
{
    SplitBanks *sb = nullptr;
    int rc = split_banks(st, C, &sb);
    if (rc) return rc;
    const size_t M = st->decay;
    Group &gm = sb->main->groups[0], &gw = sb->warm->groups[0], &g1 = st->one->groups[0];
    const size_t ld = (size_t)gm.state_ld;
    if (self_runup) {
        /* sub-chunk 0 takes the carried state; the kernel gives every other one the zero-state response of the
         * segment in front of it (no run-up launch, no scratch) */
        HIP_TRY(launch_split_seed(gm.d_state, gm.d_fval, gm.d_init, g1.d_state, g1.d_fval, g1.d_init, nullptr, C, (int)ld,
                                  0, 1, s));
        const double *saved = nullptr;
        if (in == out) {
            /* in place: a sub-chunk's last samples may be outputs by the time its successor starts, so the 1024
             * in front of every sub-chunk are set aside first (8 KiB per sub-chunk; the chooser admits in-place
             * calls only when that is the whole run-up) */
            constexpr size_t kQ = kScanSeg / 4;
            if ((rc = stream_scratch(st, (C - 1) * kQ, s))) return rc;
            HIP_TRY(hipMemcpy2DAsync(st->d_scratch, kQ * sizeof(double), in + (Lc - kQ), Lc * sizeof(double),
                                     kQ * sizeof(double), C - 1, hipMemcpyDeviceToDevice, s));
            saved = st->d_scratch;
        }
        sb->main->stream_rows = M <= kScanSeg / 4 ? 1 : 4;
        sb->main->stream_runup_src = saved;
        rc = RTFilterBank_updaten(sb->main, out, Lc, in, Lc, Lc, s);
        sb->main->stream_rows = 0;
        sb->main->stream_runup_src = nullptr;
        if (rc) return rc;
        HIP_TRY(launch_split_finish(g1.d_state, g1.d_fval, g1.d_init, gm.d_state, gm.d_fval, C, (int)ld, s));
        return RTFB_OK;
    }
    if ((rc = stream_scratch(st, (C - 1) * M, s))) return rc;

    /* run-ups: channel c-1 of `warm` filters the M samples in front of sub-chunk c from the zero state */
    HIP_TRY(launch_split_zero(gw.d_state, gw.d_init, C - 1, (int)ld, s));
    rc = RTFilterBank_updaten(sb->warm, st->d_scratch, M, in + (Lc - M), Lc, M, s);
    if (rc) return rc;

    /* start states: sub-chunk 0 takes the carried state, sub-chunk c the end state of run-up c-1 */
    HIP_TRY(launch_split_seed(gm.d_state, gm.d_fval, gm.d_init, g1.d_state, g1.d_fval, g1.d_init, gw.d_state, C, (int)ld,
                              1, 1, s));

    rc = RTFilterBank_updaten(sb->main, out, Lc, in, Lc, Lc, s);
    if (rc) return rc;

    /* the stream's state is what the last sub-chunk ended in */
    HIP_TRY(launch_split_finish(g1.d_state, g1.d_fval, g1.d_init, gm.d_state, gm.d_fval, C, (int)ld, s));
    return RTFB_OK;
}

/* Interleaved segments: the stream's 2048-sample segments dealt round-robin over one resident round of
 * workgroups, each segment started from the zero-state end vector of the one before it (dsp_internal.h:
 * Pipe16Ext).  For designs of order 9..16 whose decay length is at most one segment; the grid then sweeps ONE
 * contiguous window of the stream (the access pattern of the tiled block layout) instead of 1024 windows 8 MiB
 * apart, and no run-up launch is needed. */
constexpr size_t kStream16MinSegments = 64;
constexpr size_t kStream16MaxSegments = (size_t)1 << 19;      /* per launch: 128 MiB of granules */

bool stream16_ok(const RTFilterStream *st)
{
    if (st->kind != RTFB_KIND_IIR || !st->exact_ok || !st->decay || st->decay > kPipe16Seg) return false;
    const Group &g = st->one->groups[0];
    return g.order > kScan2MaxOrder && g.order <= kMaxFastOrder && st->one->mode == RTFB_MODE_AUTO;
}

int stream16_piece(RTFilterStream *st, double *out, const double *in, size_t nseg, size_t *grid_used, hipStream_t s)
{
    if (st->exch_segments < nseg) {
        HIP_TRY(hipStreamSynchronize(s));
        pool_free(st->d_exch);
        st->d_exch = nullptr;
        st->exch_segments = 0;
        void *p = nullptr;
        const size_t want = std::max(nseg, (size_t)4096);
        /* + room for the stamps of a diagnostic build (pipe16_kernel.hip: DSP_P16_STAMPS) */
        const size_t bytes = want * 32 * sizeof(double) + (size_t)pipe_resident_round() * 2 * 16 * sizeof(unsigned long long);
        int rc = pool_alloc(&p, bytes, st->device);
        if (rc) return rc;
        st->d_exch = static_cast<double *>(p);
        st->exch_segments = want;
        /* pool memory may hold another stream's granules with generations this one will hand out */
        HIP_TRY(hipMemsetAsync(st->d_exch, 0, bytes, s));
    }
    if (!st->h_error) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&st->h_error), 64, hipHostMallocDefault));
        *st->h_error = 0;
    }
    const Group &g1 = st->one->groups[0];
    const GroupView v = view_of(st->one, g1);
    BlockIO io;
    io.out = out;
    io.ld_out = 0;
    io.in = in;
    io.ld_in = 0;
    io.n = nseg * kPipe16Seg;
    io.aligned16 = 1;
    Pipe16Ext ext;
    ext.exch = st->d_exch;
    ext.gen_base = st->exch_gen;
    ext.error = st->h_error;
    const size_t grid = std::min(std::min((size_t)st->interleave, (size_t)pipe_resident_round()), nseg);
    HIP_TRY(launch_iir_pipe16(g1.order, v, io, ext, (int)grid, s));
    st->exch_gen += nseg;
    *grid_used = grid;
    return RTFB_OK;
}

/* A^L for the homogeneous recurrence (long double, binary powering), cached on the device per L. */
int chunk_power(RTFilterStream *st, size_t L, const double **dev)
{
    auto it = st->d_power.find(L);
    if (it != st->d_power.end()) {
        *dev = it->second;
        return RTFB_OK;
    }
    const int N = (int)st->a.size() - 1;
    typedef std::vector<long double> Mat;
    auto mul = [N](const Mat &p, const Mat &q) {
        Mat r((size_t)N * N, 0.0L);
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < N; ++k) {
                const long double pik = p[(size_t)i * N + k];
                for (int j = 0; j < N; ++j) r[(size_t)i * N + j] += pik * q[(size_t)k * N + j];
            }
        return r;
    };
    Mat P((size_t)N * N, 0.0L), R((size_t)N * N, 0.0L);
    for (int i = 0; i < N; ++i) {
        P[(size_t)i * N] = -(long double)st->a[i + 1] / (long double)st->a[0];
        if (i + 1 < N) P[(size_t)i * N + i + 1] += 1.0L;
        R[(size_t)i * N + i] = 1.0L;
    }
    for (size_t e = L; e; e >>= 1) {
        if (e & 1) R = mul(R, P);
        if (e > 1) P = mul(P, P);
    }
    std::vector<double> host((size_t)N * N);
    for (size_t e = 0; e < host.size(); ++e) host[e] = (double)R[e];
    void *p = nullptr;
    int rc = pool_alloc(&p, host.size() * sizeof(double), st->device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice));
    st->d_power[L] = static_cast<double *>(p);
    *dev = static_cast<double *>(p);
    return RTFB_OK;
}

/* One time-split piece WITHOUT a decay assumption: the chunk maps are composed exactly.
 * Pass A filters sub-chunks 0 .. C-2 from the zero state (their outputs land in `out` and are overwritten by
 * pass B) and leaves their zero-state end states v_c; s_c = A^Lc s_(c-1) + v_(c-1) gives every sub-chunk its
 * start state; pass B is the ordinary launch.  Twice the traffic of the decay shortcut, C-fold parallel where
 * the alternative is one workgroup.  in and out must not overlap; the stream must be running (not fresh). */
int stream_split_piece_exact(RTFilterStream *st, double *out, const double *in, size_t C, size_t Lc, hipStream_t s)
{
    SplitBanks *sb = nullptr;
    int rc = split_banks(st, C, &sb);
    if (rc) return rc;
    const double *d_M = nullptr;
    if ((rc = chunk_power(st, Lc, &d_M))) return rc;
    Group &gm = sb->main->groups[0], &gw = sb->warm->groups[0], &g1 = st->one->groups[0];
    const size_t ld = (size_t)gm.state_ld;
    const int N = (int)st->a.size() - 1;

    HIP_TRY(launch_split_zero(gw.d_state, gw.d_init, C - 1, (int)ld, s));
    if ((rc = RTFilterBank_updaten(sb->warm, out, Lc, in, Lc, Lc, s))) return rc;

    HIP_TRY(launch_split_seed(gm.d_state, gm.d_fval, gm.d_init, g1.d_state, g1.d_fval, g1.d_init, gw.d_state, C, (int)ld,
                              0, 0, s));
    HIP_TRY(launch_compose_states(gm.d_state, ld, g1.d_state, gw.d_state, ld, d_M, N, (int)C, 1, s));
    if (getenv("DSP_AMD_DEBUG_EXACT")) {
        (void)hipDeviceSynchronize();
        std::vector<double> hv((C - 1) * ld), hs(C * ld), hm((size_t)N * N);
        std::vector<int32_t> hi(C);
        (void)hipMemcpy(hv.data(), gw.d_state, hv.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hs.data(), gm.d_state, hs.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hm.data(), d_M, hm.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hi.data(), gm.d_init, hi.size() * 4, hipMemcpyDeviceToHost);
        fprintf(stderr, "EXACT C=%zu Lc=%zu M=[%g %g %g %g]\n", C, Lc, hm[0], hm[1], hm[2], hm[3]);
        for (size_t c = 40; c < 47 && c < C - 1; ++c)
            fprintf(stderr, "  v[%zu] = %.17g %.17g %.17g | s[%zu] = %.17g %.17g %.17g init %d\n", c, hv[c * ld], hv[c * ld + 1],
                    hv[c * ld + 2], c, hs[c * ld], hs[c * ld + 1], hs[c * ld + 2], hi[c]);
    }

    if ((rc = RTFilterBank_updaten(sb->main, out, Lc, in, Lc, Lc, s))) return rc;

    HIP_TRY(launch_split_finish(g1.d_state, g1.d_fval, g1.d_init, gm.d_state, gm.d_fval, C, (int)ld, s));
    return RTFB_OK;
}

int stream_create(RTFilterStream **stream, RTFilterStream *st)
{
    int rc = RTFB_OK;
    if (st->kind == RTFB_KIND_IIR && !st->one->groups.empty() && st->one->groups[0].cls == kIirFast &&
        st->one->designs[0].scan_safe() && iir_scan_supported(st->one->groups[0].order)) {
        st->decay = decay_length(st->a);
        if (st->decay) rc = one_channel_bank(&st->tail, st);
        st->exact_ok = true;
    }
    if (rc) {
        RTFilterStream_destroy(st);
        return rc;
    }
    *stream = st;
    return RTFB_OK;
}

}  // namespace

extern "C" {

int RTFilterStream_from_design(RTFilterStream **stream, const RTFilterDesign *design, int device)
{
    if (!stream || !design) return fail(RTFB_EINVAL, "RTFilterStream_from_design: NULL argument");
    *stream = nullptr;
    int rc = validate_design(*design);
    if (rc) return rc;
    if ((rc = check_device(device))) return rc;
    RTFilterStream *st = new RTFilterStream;
    st->device = device;
    st->kind = design->kind;
    st->b.assign(design->b, design->b + design->nb);
    if (design->kind == RTFB_KIND_IIR) st->a.assign(design->a, design->a + design->na);
    if ((rc = one_channel_bank(&st->one, st))) {
        delete st;
        return rc;
    }
    return stream_create(stream, st);
}

int RTFilterStream_from_filter(RTFilterStream **stream, RTFilter *filter, int device)
{
    if (!stream || !filter) return fail(RTFB_EINVAL, "RTFilterStream_from_filter: NULL argument");
    *stream = nullptr;
    RTFilterBank *one = nullptr;
    RTFilter *const list[1] = {filter};
    int rc = RTFilterBank_from_filters(&one, list, 1, device);
    if (rc) return rc;
    RTFilterStream *st = new RTFilterStream;
    st->device = device;
    st->one = one;
    const Design &d = one->designs[0];
    st->kind = d.kind;
    st->b = d.b;
    st->a = d.a;
    return stream_create(stream, st);
}

void RTFilterStream_destroy(RTFilterStream *st)
{
    if (!st) return;
    for (auto &kv : st->split) {
        RTFilterBank_destroy(kv.second.main);
        RTFilterBank_destroy(kv.second.warm);
    }
    RTFilterBank_destroy(st->tail);
    RTFilterBank_destroy(st->one);             /* synchronises the device */
    pool_free(st->d_scratch);
    pool_free(st->d_split_mine);
    pool_free(st->d_split_all);
    pool_free(st->d_exch);
    if (st->h_error) (void)hipHostFree(st->h_error);
    for (auto &kv : st->d_power) pool_free(kv.second);
    host_pipe_release(st->host);
    delete st;
}

size_t RTFilterStream_decay_length(const RTFilterStream *st) { return st ? st->decay : 0; }
size_t RTFilterStream_last_split(const RTFilterStream *st) { return st ? st->last_split : 0; }

int RTFilterStream_updaten(RTFilterStream *st, double *out, const double *in, size_t n, void *hip_stream)
{
    if (!st || !out || !in) return fail(RTFB_EINVAL, "RTFilterStream_updaten: NULL argument");
    if (n > (size_t)1 << 40) return fail(RTFB_ESIZE, "RTFilterStream_updaten: n too large");
    DeviceScope ds(st->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    st->last_split = 1;
    bool first = true;
    const bool overlap = in < out + n && out < in + n;
    if (st->h_error && *st->h_error)
        return fail(RTFB_EDEVICE, "RTFilterStream_updaten: an earlier call's workgroups waited in vain for each other (" +
                                      std::to_string(*st->h_error) + "): the grid was not resident at once; the "
                                      "stream's state is undefined");
    const bool whole16 = (reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16 == 0 &&
                         (!overlap || in == out);
    while (n) {
        if (whole16 && st->interleave > 0 && n >= kStream16MinSegments * kPipe16Seg && stream16_ok(st)) {
            const size_t nseg = std::min(n / kPipe16Seg, kStream16MaxSegments);
            size_t grid = 0;
            int rc = stream16_piece(st, out, in, nseg, &grid, s);
            if (rc) return rc;
            if (first) st->last_split = grid;
            first = false;
            out += nseg * kPipe16Seg;
            in += nseg * kPipe16Seg;
            n -= nseg * kPipe16Seg;
            continue;
        }
        /* sub-chunk: at least 4 decay lengths (the run-up costs M/Lc extra) and 8192 samples */
        size_t lmin = std::max<size_t>(4 * st->decay, 8192);
        lmin = (lmin + kStreamGrain - 1) / kStreamGrain * kStreamGrain;
        size_t C = st->decay ? std::min(kStreamMaxSplit, n / lmin) : 0;
        /* orders 9..16: the kernel does the run-ups of the sub-chunks itself (no run-up launch: 15-20 us of a call;
         * it reads the samples in front of a sub-chunk -- in place, where they may already hold outputs, from a
         * copy set aside beforehand, which is kept to run-ups of 1024 samples), and long streams are cut into more,
         * shorter sub-chunks */
        bool self_runup = false;
        size_t grain = kStreamGrain;
        if (st->decay && st->decay <= kScanSeg && whole16 && (!overlap || st->decay <= kScanSeg / 4) &&
            scan16_serves(st->one, kScanSeg) && st->one->groups[0].scan_safe) {
            if (n / std::max(lmin, kStreamChunkTarget) > kStreamMaxSplit)
                C = std::min(kStreamMaxSplitShort, n / std::max(lmin, kStreamChunkTarget));
            if (C >= kStreamMinSplit) {                 /* (n / C >= lmin >= two segments) */
                self_runup = true;
                grain = kScanSeg;
            }
        }
        bool exact = false;
        if (C < kStreamMinSplit && st->exact_ok && !overlap) {
            /* no (usable) decay: compose the chunk maps exactly instead of falling to one workgroup.  A stream
             * that may still be fresh first sends one segment through the sequential path, which runs the
             * initialiser on sample 0 (the composition needs affine maps). */
            if (st->maybe_fresh && n > kStreamGrain) {
                int rc = RTFilterBank_updaten(st->one, out, kStreamGrain, in, kStreamGrain, kStreamGrain, s);
                if (rc) return rc;
                st->maybe_fresh = false;
                out += kStreamGrain;
                in += kStreamGrain;
                n -= kStreamGrain;
            }
            const size_t Ce = std::min(kStreamMaxSplit, n / kStreamExactChunk);
            if (Ce >= kStreamMinSplit && !st->maybe_fresh) {
                C = Ce;
                exact = true;
            }
        }
        if (C < kStreamMinSplit) {                    /* sequential: the one-channel bank itself */
            int rc = RTFilterBank_updaten(st->one, out, n, in, n, n, s);
            if (rc) return rc;
            break;
        }
        const size_t Lc = n / C / grain * grain;
        if (first) st->last_split = C;
        first = false;
        int rc = exact ? stream_split_piece_exact(st, out, in, C, Lc, s)
                       : stream_split_piece(st, out, in, C, Lc, self_runup, s);
        if (rc) return rc;
        out += C * Lc;
        in += C * Lc;
        n -= C * Lc;
    }
    st->maybe_fresh = false;
    return RTFB_OK;
}

int RTFilterStream_updaten_host(RTFilterStream *st, double *out, const double *in, size_t n)
{
    if (!st || !out || !in) return fail(RTFB_EINVAL, "RTFilterStream_updaten_host: NULL argument");
    if (!n) return RTFB_OK;
    DeviceScope ds(st->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    const size_t nt = std::min(n, kHostChunkBytes / sizeof(double));      /* 2^23 samples per chunk */
    int rc = host_pipe_prepare(st->host, st->device, std::max(nt * sizeof(double), (size_t)4096));
    if (rc) return rc;
    return host_pipe_run(st->host, 1, n, in, n, out, n, nt,
                         [&](double *d_out, const double *d_in, size_t len, hipStream_t s) {
                             return RTFilterStream_updaten(st, d_out, d_in, len, s);
                         });
}

int RTFilterStream_reset(RTFilterStream *st, void *hip_stream)
{
    if (!st) return fail(RTFB_EINVAL, "NULL stream");
    st->maybe_fresh = true;
    return RTFilterBank_reset(st->one, hip_stream);
}

int RTFilterStream_sync_to_host(RTFilterStream *st, void *hip_stream)
{
    if (!st) return fail(RTFB_EINVAL, "NULL stream");
    return RTFilterBank_sync_to_host(st->one, hip_stream);
}

size_t RTFilterStream_pack_len(const RTFilterStream *st)
{
    return st ? (size_t)st->one->groups[0].state_ld + 2 : 0;
}

int RTFilterStream_export_state(RTFilterStream *st, double *pack_dev, void *hip_stream)
{
    if (!st || !pack_dev) return fail(RTFB_EINVAL, "RTFilterStream_export_state: NULL argument");
    DeviceScope ds(st->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    const Group &g = st->one->groups[0];
    HIP_TRY(launch_state_pack(pack_dev, g.d_state, g.d_fval, g.d_init, g.state_ld,
                              static_cast<hipStream_t>(hip_stream)));
    return RTFB_OK;
}

int RTFilterStream_import_state(RTFilterStream *st, const double *pack_dev, void *hip_stream)
{
    if (!st || !pack_dev) return fail(RTFB_EINVAL, "RTFilterStream_import_state: NULL argument");
    DeviceScope ds(st->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    const Group &g = st->one->groups[0];
    HIP_TRY(launch_state_unpack(pack_dev, g.d_state, g.d_fval, g.d_init, g.state_ld,
                                static_cast<hipStream_t>(hip_stream)));
    st->maybe_fresh = true;                  /* the imported `initialized` is not known on the host */
    return RTFB_OK;
}

int RTFilterStream_exact_split(const RTFilterStream *st) { return st && st->exact_ok ? 1 : 0; }

int RTFilterStream_set_interleave(RTFilterStream *st, int workgroups)
{
    if (!st) return fail(RTFB_EINVAL, "NULL stream");
    st->interleave = workgroups;
    return RTFB_OK;
}

int RTFilterStream_interleaved(const RTFilterStream *st) { return st && stream16_ok(st) ? 1 : 0; }

/* diagnostics (tools/stream16_race.py): the granule buffer of the last interleaved call */
__attribute__((visibility("default"))) int RTFilterStream_debug_exchange(const RTFilterStream *st, double **ptr, size_t *segments,
                                                                         unsigned long long *gen_end)
{
    if (!st) return RTFB_EINVAL;
    *ptr = st->d_exch;
    *segments = st->exch_segments;
    *gen_end = st->exch_gen;
    return RTFB_OK;
}

int RTFilterStream_zero_state(RTFilterStream *st, void *hip_stream)
{
    if (!st) return fail(RTFB_EINVAL, "NULL stream");
    DeviceScope ds(st->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const Group &g = st->one->groups[0];
    HIP_TRY(hipMemsetAsync(g.d_state, 0, (size_t)g.state_ld * sizeof(double), s));
    HIP_TRY(hipMemsetAsync(g.d_fval, 0, sizeof(double), s));
    HIP_TRY(hipMemsetAsync(g.d_init, 0, sizeof(int32_t), s));
    st->maybe_fresh = false;
    return RTFB_OK;
}

int RTFilterStream_compose_state(RTFilterStream *st, const double *packs_dev, size_t stride, size_t count,
                                 size_t chunk_len, void *hip_stream)
{
    if (!st || !packs_dev || !count) return fail(RTFB_EINVAL, "RTFilterStream_compose_state: NULL or empty argument");
    if (!st->exact_ok) return fail(RTFB_EUNSUPPORTED, "RTFilterStream_compose_state: chunk maps of this design are not composed");
    const Group &g = st->one->groups[0];
    if (stride < (size_t)g.state_ld + 2 || count > (size_t)1 << 20)
        return fail(RTFB_EINVAL, "RTFilterStream_compose_state: stride smaller than a hand-off vector");
    DeviceScope ds(st->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const double *d_M = nullptr;
    int rc = chunk_power(st, chunk_len, &d_M);
    if (rc) return rc;
    const int N = (int)st->a.size() - 1;
    HIP_TRY(launch_compose_states(g.d_state, 0, packs_dev, packs_dev + stride, stride, d_M, N, (int)count, 0, s));
    HIP_TRY(hipMemcpyAsync(g.d_fval, packs_dev + (count - 1) * stride + g.state_ld, sizeof(double),
                           hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemsetAsync(g.d_init, 0, sizeof(int32_t), s));
    st->maybe_fresh = false;
    return RTFB_OK;
}

int RTFilterStream_tail_state(RTFilterStream *st, const double *in, size_t n, double *pack_dev,
                              void *hip_stream)
{
    if (!st || !in || !pack_dev) return fail(RTFB_EINVAL, "RTFilterStream_tail_state: NULL argument");
    if (!st->decay || !st->tail)
        return fail(RTFB_EUNSUPPORTED, "RTFilterStream_tail_state: the design does not decay; chain the true end states");
    if (n < st->decay)
        return fail(RTFB_EUNSUPPORTED, "RTFilterStream_tail_state: chunk shorter than the decay length");
    DeviceScope ds(st->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const size_t M = st->decay;
    const Group &g = st->tail->groups[0];
    if (M <= kScanSeg && n >= kScanSeg && scan16_serves(st->tail, kScanSeg) && g.scan_safe &&
        reinterpret_cast<uintptr_t>(in + n) % 16 == 0) {
        /* orders 9..16: the scan kernel's run-up alone, on the samples in front of in + n (no samples of its own) */
        BlockIO io;
        io.out = nullptr;
        io.ld_out = 0;
        io.in = in + n;
        io.ld_in = 0;
        io.n = 0;
        io.aligned16 = 1;
        io.runup = (M <= kScanSeg / 4 ? 1 : 4) | 8;
        HIP_TRY(launch_iir_scan(g.order, view_of(st->tail, g), io, s));
        HIP_TRY(launch_state_pack(pack_dev, g.d_state, g.d_fval, g.d_init, g.state_ld, s));
        return RTFB_OK;
    }
    int rc = stream_scratch(st, std::max<size_t>(M, 1), s);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(g.d_state, 0, (size_t)g.state_ld * sizeof(double), s));
    HIP_TRY(hipMemsetAsync(g.d_init, 0, sizeof(int32_t), s));
    if ((rc = RTFilterBank_updaten(st->tail, st->d_scratch, M, in + (n - M), M, M, s))) return rc;
    HIP_TRY(launch_state_pack(pack_dev, g.d_state, g.d_fval, g.d_init, g.state_ld, s));
    return RTFB_OK;
}

/* ---- the time split across GPUs, driven from C ------------------------------------------------------------- */

namespace {

int split_buffers(RTFilterStream *st, size_t world, hipStream_t s)
{
    const size_t plen = RTFilterStream_pack_len(st);
    if (st->split_world >= world && st->d_split_mine) return RTFB_OK;
    HIP_TRY(hipStreamSynchronize(s));
    pool_free(st->d_split_mine);
    pool_free(st->d_split_all);
    st->d_split_mine = st->d_split_all = nullptr;
    st->split_world = 0;
    void *p = nullptr;
    int rc = pool_alloc(&p, 2 * plen * sizeof(double), st->device);
    if (rc) return rc;
    st->d_split_mine = static_cast<double *>(p);
    if ((rc = pool_alloc(&p, world * 2 * plen * sizeof(double), st->device))) return rc;
    st->d_split_all = static_cast<double *>(p);
    st->split_world = world;
    return RTFB_OK;
}

}  // namespace

int RTFilterStream_split_updaten(RTFilterStream *st, double *out, const double *in, const size_t *lens, int rank,
                                 int world, RTFilterStream_allgather_fn allgather, void *ctx, void *hip_stream)
{
    if (!st || !lens || world < 1 || rank < 0 || rank >= world)
        return fail(RTFB_EINVAL, "RTFilterStream_split_updaten: NULL argument or rank outside the world");
    const size_t n = lens[rank];
    if (n && (!out || !in)) return fail(RTFB_EINVAL, "RTFilterStream_split_updaten: NULL buffer");
    if (world == 1) return RTFilterStream_updaten(st, out, in, n, hip_stream);
    if (!allgather) return fail(RTFB_EINVAL, "RTFilterStream_split_updaten: no all-gather");
    DeviceScope ds(st->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    /* the hand-off mode follows from the partition alone, so every rank decides alike without a message */
    bool feeders_decay = st->decay != 0, feeders_busy = true, middle_equal = true;
    for (int r = 0; r + 1 < world; ++r) {
        if (lens[r] < st->decay) feeders_decay = false;
        if (lens[r] == 0) feeders_busy = false;
        if (r >= 1 && lens[r] != lens[1]) middle_equal = false;
    }
    const bool tail = feeders_decay && st->tail;
    const bool exact = !tail && st->exact_ok && feeders_busy && middle_equal;
    if (!tail && !exact)
        return fail(RTFB_EUNSUPPORTED, "RTFilterStream_split_updaten: neither the decay tail nor composed chunk maps serve "
                                       "this design and partition; chain the ranks with _export_state / _import_state");
    int rc = split_buffers(st, (size_t)world, s);
    if (rc) return rc;
    const size_t plen = RTFilterStream_pack_len(st), row = 2 * plen;
    double *mine = st->d_split_mine, *all = st->d_split_all;
    auto gather = [&]() {
        const int grc = allgather(ctx, mine, all, row, hip_stream);
        return grc ? fail(RTFB_EDEVICE, "RTFilterStream_split_updaten: the caller's all-gather failed with " +
                                            std::to_string(grc))
                   : RTFB_OK;
    };
    if (tail) {
        /* row [0]: the hand-off vector of my chunk (zero on the last rank), [1]: my state at entry -- the last
         * rank's is the stream's end state of the call before and rides back to rank 0 here */
        if ((rc = RTFilterStream_export_state(st, mine + plen, s))) return rc;
        if (rank + 1 < world) {
            if ((rc = RTFilterStream_tail_state(st, in, n, mine, s))) return rc;
        } else
            HIP_TRY(hipMemsetAsync(mine, 0, plen * sizeof(double), s));
        if ((rc = gather())) return rc;
        if (rank == 0) {
            if (st->split_carry_remote && (rc = RTFilterStream_import_state(st, all + (size_t)(world - 1) * row + plen, s)))
                return rc;
        } else if ((rc = RTFilterStream_import_state(st, all + (size_t)(rank - 1) * row, s)))
            return rc;
        if (n && (rc = RTFilterStream_updaten(st, out, in, n, s))) return rc;     /* (a last rank may hold nothing) */
    } else {
        if (st->split_carry_remote) {                       /* rank 0 needs its start state before its pass */
            if ((rc = RTFilterStream_export_state(st, mine + plen, s)) || (rc = gather())) return rc;
            if (rank == 0 && (rc = RTFilterStream_import_state(st, all + (size_t)(world - 1) * row + plen, s))) return rc;
        }
        if (rank == 0) {
            if ((rc = RTFilterStream_updaten(st, out, in, n, s)) || (rc = RTFilterStream_export_state(st, mine, s)))
                return rc;
        } else if (rank + 1 < world) {
            /* zero-state pass: only its end state is wanted; its outputs must not land on inputs that the true
             * pass reads again */
            double *first_out = out;
            if (in < out + n && out < in + n) {
                if ((rc = stream_scratch(st, n, s))) return rc;
                first_out = st->d_scratch;
            }
            if ((rc = RTFilterStream_zero_state(st, s)) || (rc = RTFilterStream_updaten(st, first_out, in, n, s)) ||
                (rc = RTFilterStream_export_state(st, mine, s)))
                return rc;
        } else
            HIP_TRY(hipMemsetAsync(mine, 0, plen * sizeof(double), s));
        if ((rc = gather())) return rc;
        if (rank > 0) {
            if ((rc = RTFilterStream_compose_state(st, all, row, (size_t)rank, lens[1], s)) ||
                (n && (rc = RTFilterStream_updaten(st, out, in, n, s))))
                return rc;
        }
    }
    st->split_carry_remote = true;
    return RTFB_OK;
}

int RTFilterStream_split_end_state(RTFilterStream *st, int rank, int world, RTFilterStream_allgather_fn allgather,
                                   void *ctx, void *hip_stream)
{
    if (!st || world < 1 || rank < 0 || rank >= world) return fail(RTFB_EINVAL, "RTFilterStream_split_end_state: bad argument");
    if (world == 1 || !st->split_carry_remote) return RTFB_OK;
    if (!allgather) return fail(RTFB_EINVAL, "RTFilterStream_split_end_state: no all-gather");
    DeviceScope ds(st->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    int rc = split_buffers(st, (size_t)world, s);
    if (rc) return rc;
    const size_t plen = RTFilterStream_pack_len(st), row = 2 * plen;
    if ((rc = RTFilterStream_export_state(st, st->d_split_mine + plen, s))) return rc;
    HIP_TRY(hipMemsetAsync(st->d_split_mine, 0, plen * sizeof(double), s));
    if (allgather(ctx, st->d_split_mine, st->d_split_all, row, hip_stream))
        return fail(RTFB_EDEVICE, "RTFilterStream_split_end_state: the caller's all-gather failed");
    if ((rc = RTFilterStream_import_state(st, st->d_split_all + (size_t)(world - 1) * row + plen, s))) return rc;
    st->split_carry_remote = false;
    return RTFB_OK;
}

/* RCCL as the collective.  The library does not link RCCL: the one routine it needs is looked up in the process
 * (a caller that holds an ncclComm_t has it loaded) or in librccl.so.1 on first use. */
namespace {

typedef int (*nccl_allgather_t)(const void *, void *, size_t, int, void *, hipStream_t);

nccl_allgather_t rccl_allgather()
{
    static nccl_allgather_t fn = [] {
        void *sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (!sym) {
            void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (lib) sym = dlsym(lib, "ncclAllGather");
        }
        return reinterpret_cast<nccl_allgather_t>(sym);
    }();
    return fn;
}

int rccl_gather_cb(void *ctx, const double *send_dev, double *recv_dev, size_t count, void *hip_stream)
{
    constexpr int kNcclDouble = 8;                          /* rccl.h: ncclFloat64 = ncclDouble = 8 */
    return rccl_allgather()(send_dev, recv_dev, count, kNcclDouble, ctx, static_cast<hipStream_t>(hip_stream));
}

}  // namespace

/* diagnostics (tests/test_gpu_rccl_one_rank.py): the adapter's one RCCL call on a communicator of the caller's.  Not
 * declared in the headers. */
EXPORT int dsp_debug_rccl_allgather(void *nccl_comm, const double *send_dev, double *recv_dev, size_t count,
                                    void *hip_stream)
{
    if (!nccl_comm || !rccl_allgather()) return fail(RTFB_EDEVICE, "no communicator, or ncclAllGather not found");
    return rccl_gather_cb(nccl_comm, send_dev, recv_dev, count, hip_stream);
}

int RTFilterStream_split_updaten_rccl(RTFilterStream *st, double *out, const double *in, const size_t *lens, int rank,
                                      int world, void *nccl_comm, void *hip_stream)
{
    if (world > 1 && (!nccl_comm || !rccl_allgather()))
        return fail(RTFB_EDEVICE, "RTFilterStream_split_updaten_rccl: no communicator, or ncclAllGather not found "
                                  "(librccl.so.1)");
    return RTFilterStream_split_updaten(st, out, in, lens, rank, world, rccl_gather_cb, nccl_comm, hip_stream);
}

int RTFilterStream_split_end_state_rccl(RTFilterStream *st, int rank, int world, void *nccl_comm, void *hip_stream)
{
    if (world > 1 && (!nccl_comm || !rccl_allgather()))
        return fail(RTFB_EDEVICE, "RTFilterStream_split_end_state_rccl: no communicator, or ncclAllGather not found");
    return RTFilterStream_split_end_state(st, rank, world, rccl_gather_cb, nccl_comm, hip_stream);
}

int RTFilterStream_get_state(RTFilterStream *st, double *state, double *fval, int32_t *init, void *hip_stream)
{
    if (!st) return fail(RTFB_EINVAL, "NULL stream");
    return RTFilterBank_get_state(st->one, state, fval, init, hip_stream);
}

int RTFilterStream_set_state(RTFilterStream *st, const double *state, const double *fval, const int32_t *init,
                             void *hip_stream)
{
    if (!st) return fail(RTFB_EINVAL, "NULL stream");
    st->maybe_fresh = true;
    return RTFilterBank_set_state(st->one, state, fval, init, hip_stream);
}

}  /* extern "C" */

/* ======================================================================== */
/* RTFilterBankFeed: per-sample producers coalesced into device blocks         */
/* (the reference's per-sample entry is one FFI        */
/* crossing and one filter per call: src/RTFilter.c:98-108, python/dsp.py:6-13,35-38) */

struct RTFilterBankFeed {
    RTFilterBank *bank = nullptr;
    int device = 0;                          /* the bank's device, kept here: the feed may outlive the bank's handle */
    size_t nch = 0, block = 0;               /* frames per device block */
    hipStream_t s = nullptr;
    double *h_in[2] = {nullptr, nullptr}, *h_out[2] = {nullptr, nullptr};      /* pinned, time-major [block][nch] */
    double *d_tm = nullptr, *d_cm = nullptr;                                   /* device: time-major, channel-major */
    hipEvent_t done[2] = {nullptr, nullptr};
    size_t fill = 0;                         /* frames in the buffer being filled */
    int fill_slot = 0;
    /* blocks submitted and not yet popped, oldest first (at most 2): slot, frames, frames already popped */
    struct InFlight { int slot; size_t frames, popped; };
    std::deque<InFlight> queue;
};

namespace {

void feed_release(RTFilterBankFeed *f)
{
    DeviceScope ds(f->device);               /* not f->bank->device: the bank may already be destroyed */
    if (f->s) (void)hipStreamSynchronize(f->s);
    for (int k = 0; k < 2; ++k) {
        if (f->h_in[k]) (void)hipHostFree(f->h_in[k]);
        if (f->h_out[k]) (void)hipHostFree(f->h_out[k]);
        if (f->done[k]) (void)hipEventDestroy(f->done[k]);
    }
    pool_free(f->d_tm);
    pool_free(f->d_cm);
    if (f->s) (void)hipStreamDestroy(f->s);
}

/* the block in `fill_slot` goes to the device: H2D, frames -> rows, filter in place, rows -> frames, D2H */
int feed_submit(RTFilterBankFeed *f)
{
    const size_t frames = f->fill;
    if (!frames) return RTFB_OK;
    const int slot = f->fill_slot;
    const size_t bytes = frames * f->nch * sizeof(double);
    HIP_TRY(hipMemcpyAsync(f->d_tm, f->h_in[slot], bytes, hipMemcpyHostToDevice, f->s));
    HIP_TRY(launch_transpose(f->d_cm, f->block, f->d_tm, f->nch, frames, f->nch, f->s));
    int rc = RTFilterBank_updaten(f->bank, f->d_cm, f->block, f->d_cm, f->block, frames, f->s);
    if (rc) return rc;
    HIP_TRY(launch_transpose(f->d_tm, f->nch, f->d_cm, f->block, f->nch, frames, f->s));
    HIP_TRY(hipMemcpyAsync(f->h_out[slot], f->d_tm, bytes, hipMemcpyDeviceToHost, f->s));
    HIP_TRY(hipEventRecord(f->done[slot], f->s));
    f->queue.push_back({slot, frames, 0});
    f->fill = 0;
    f->fill_slot ^= 1;
    return RTFB_OK;
}

}  // namespace

extern "C" {

int RTFilterBankFeed_create(RTFilterBankFeed **feed, RTFilterBank *bank, size_t block_frames)
{
    if (!feed || !bank || !block_frames) return fail(RTFB_EINVAL, "RTFilterBankFeed_create: NULL or empty argument");
    *feed = nullptr;
    if (block_frames > ((size_t)1 << 24) || bank->nch * block_frames > ((size_t)1 << 28))
        return fail(RTFB_ESIZE, "RTFilterBankFeed_create: channels x block_frames above 2^28 doubles");
    DeviceScope ds(bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    RTFilterBankFeed *f = new RTFilterBankFeed;
    f->bank = bank;
    f->device = bank->device;
    f->nch = bank->nch;
    f->block = block_frames;
    const size_t bytes = f->nch * f->block * sizeof(double);
    int rc = RTFB_OK;
    auto hipok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) rc = fail_hip(e, what);
        return e == hipSuccess;
    };
    hipok(hipStreamCreateWithFlags(&f->s, hipStreamNonBlocking), "hipStreamCreate");
    for (int k = 0; k < 2 && !rc; ++k)
        hipok(hipHostMalloc(reinterpret_cast<void **>(&f->h_in[k]), bytes, hipHostMallocDefault), "hipHostMalloc") &&
            hipok(hipHostMalloc(reinterpret_cast<void **>(&f->h_out[k]), bytes, hipHostMallocDefault), "hipHostMalloc") &&
            hipok(hipEventCreateWithFlags(&f->done[k], hipEventDisableTiming), "hipEventCreate");
    void *p = nullptr;
    if (!rc && !(rc = pool_alloc(&p, bytes, bank->device))) f->d_tm = static_cast<double *>(p);
    if (!rc && !(rc = pool_alloc(&p, bytes, bank->device))) f->d_cm = static_cast<double *>(p);
    if (rc) {
        const std::string keep = t_last_error;
        feed_release(f);
        delete f;
        t_last_error = keep;
        return rc;
    }
    *feed = f;
    return RTFB_OK;
}

void RTFilterBankFeed_destroy(RTFilterBankFeed *feed)
{
    if (!feed) return;
    feed_release(feed);
    delete feed;
}

int RTFilterBankFeed_push(RTFilterBankFeed *f, const double *frames, size_t nframes, size_t *accepted)
{
    if (accepted) *accepted = 0;
    if (!f || (!frames && nframes)) return fail(RTFB_EINVAL, "RTFilterBankFeed_push: NULL argument");
    DeviceScope ds(f->bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    while (nframes) {
        if (f->fill == 0 && f->queue.size() == 2) {          /* both output buffers hold unpopped frames */
            if (accepted) return RTFB_OK;                    /* the caller sees how far it got and pops */
            return fail(RTFB_ESIZE, "RTFilterBankFeed_push: two blocks are waiting to be popped; pop first");
        }
        const size_t take = std::min(nframes, f->block - f->fill);
        std::memcpy(f->h_in[f->fill_slot] + f->fill * f->nch, frames, take * f->nch * sizeof(double));
        f->fill += take;
        frames += take * f->nch;
        nframes -= take;
        if (accepted) *accepted += take;
        if (f->fill == f->block) {
            int rc = feed_submit(f);
            if (rc) return rc;
        }
    }
    return RTFB_OK;
}

int RTFilterBankFeed_flush(RTFilterBankFeed *f)
{
    if (!f) return fail(RTFB_EINVAL, "NULL feed");
    DeviceScope ds(f->bank->device);
    if (!ds.ok) return fail(RTFB_EDEVICE, "hipSetDevice failed");
    if (f->fill && f->queue.size() == 2)
        return fail(RTFB_ESIZE, "RTFilterBankFeed_flush: two blocks are waiting to be popped; pop first");
    int rc = feed_submit(f);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(f->s));
    return RTFB_OK;
}

/* Blocks until the oldest submitted block has come back (nothing to wait for: returns at once). */
int RTFilterBankFeed_wait(RTFilterBankFeed *f)
{
    if (!f) return fail(RTFB_EINVAL, "NULL feed");
    if (f->queue.empty()) return RTFB_OK;
    HIP_TRY(hipEventSynchronize(f->done[f->queue.front().slot]));
    return RTFB_OK;
}

/* Submitted blocks that have not been popped completely (0..2). */
size_t RTFilterBankFeed_pending(RTFilterBankFeed *f) { return f ? f->queue.size() : 0; }

size_t RTFilterBankFeed_available(RTFilterBankFeed *f)
{
    if (!f) return 0;
    size_t n = 0;
    for (const RTFilterBankFeed::InFlight &b : f->queue) {
        if (hipEventQuery(f->done[b.slot]) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        n += b.frames - b.popped;
    }
    return n;
}

int RTFilterBankFeed_pop(RTFilterBankFeed *f, double *frames, size_t max_frames, size_t *got)
{
    if (!f || !got || (!frames && max_frames)) return fail(RTFB_EINVAL, "RTFilterBankFeed_pop: NULL argument");
    *got = 0;
    while (max_frames && !f->queue.empty()) {
        RTFilterBankFeed::InFlight &b = f->queue.front();
        if (hipEventQuery(f->done[b.slot]) != hipSuccess) {
            (void)hipGetLastError();
            break;                               /* not there yet: non-blocking */
        }
        const size_t take = std::min(max_frames, b.frames - b.popped);
        std::memcpy(frames, f->h_out[b.slot] + b.popped * f->nch, take * f->nch * sizeof(double));
        frames += take * f->nch;
        b.popped += take;
        *got += take;
        max_frames -= take;
        if (b.popped == b.frames) f->queue.pop_front();
    }
    return RTFB_OK;
}

}  /* extern "C" */
