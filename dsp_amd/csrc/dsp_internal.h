/* dsp_internal.h -- shared between bank.hip (host-side bank manager) and
 * kernels.hip (gfx950 kernels + their launchers).  Not installed. */
#ifndef DSP_INTERNAL_H
#define DSP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dsp {

constexpr int kWave = 64;            /* CDNA wavefront */
constexpr int kMaxFastOrder = 16;    /* IIR orders with unrolled kernels (na == nb == N+1) */
constexpr int kMaxGenericTaps = 64;  /* na, nb limit of the generic IIR kernel */
constexpr int kScanLevels = 6;       /* log2(kWave): Kogge-Stone steps across a wavefront */

/* Sub-chunk (samples per lane per tile) of each kernel family.  A tile is
 * 64 rows x T doubles staged through LDS with one 16-byte pad per row. */
constexpr int kSeqT = 32;            /* sequential kernel: rows are channels          */
constexpr int kScanT = 64;           /* scan kernel: rows are consecutive time chunks */
constexpr int kScan2T = 32;          /* two-wave scan kernel: samples per lane                */
constexpr int kScan2Rows = 128;      /* ... lanes (= time chunks) per segment                 */
constexpr int kScan2Levels = 7;      /* log2(kScan2Rows)                                      */
constexpr int kScan2MaxOrder = 8;    /* orders served by the two-wave kernel (LDS budget)     */
constexpr int kFirT = 32;            /* FIR kernel: outputs per lane                  */
constexpr int kFirR = 8;             /* FIR kernel: outputs accumulated together      */
constexpr int kFirSeg = kWave * kFirT;   /* outputs per wavefront                    */
constexpr int kMaxFirTaps = 4096;
constexpr int kFirMfmaMinTaps = 2;   /* the persistent matrix-core kernel streams at ~4.9 TB/s from 2 taps up; the bit-exact VALU kernel falls from 4.6 (2 taps) to 4.1 (16): profiles/r2_fir_sweep.txt */

/* One launch group: channels of the bank that share a kernel. All pointers are
 * device pointers; arrays are indexed by the group-local channel index g. */
struct GroupView {
    const int32_t *chan;      /* global channel id per g, or nullptr: channel = chan0 + g */
    int32_t        chan0;
    int32_t        count;
    const int32_t *design;    /* design index per g                                       */
    double        *state;     /* count x state_ld                                         */
    double        *state_out; /* FIR only: the block's end state is written here (ping-pong) */
    int32_t        state_ld;
    int32_t       *aux;       /* FIR only: 2 ints per g written by the prologue: first regular
                                 sample t0 (1 if sample 0 went to stable_init) and the stream
                                 index of the block's sample 0 while a partial_init ramp runs */
    int32_t        kmax;      /* FIR only: largest tap count in the group                 */
    int32_t        unit_a0;   /* IIR fast groups: every design has a[0] == 1.0 exactly    */
    int32_t        fir_mfma;  /* FIR: outputs come from the Toeplitz-tile MFMA kernel (exact head kernel
                                 only for channels inside a partial_init ramp)                  */
    int32_t        has_partial;   /* FIR: some design of the group is RTFB_KIND_FIR_PARTIAL     */
    int32_t        single_design; /* design index shared by every channel of the group, or -1    */
    double        *fval;      /* RTFilter.filtered_value per g                            */
    int32_t       *init;      /* RTFilter.initialized per g                               */
    const int32_t *coef_off;  /* per design: offset of b[0] in coefs (a follows b)        */
    const int32_t *pow_off;   /* per design: offset of the kScanLevels N x N matrices     */
    const int32_t *pow2_off;  /* per design: offset of the kScan2Levels matrices A^(32 2^k) */
    const int32_t *levels2;   /* per design: combine levels that still matter (decay cut)  */
    const int32_t *ttab_off;  /* per design: offset in pows of the 64 row propagators A^(32 j),
                                 j = 0..63, stored [element][j] (pipelined kernel); -1 if not built */
    const int32_t *gtab_off;  /* per design: offset in pows of G[n][i] = (A^(31-n) B)_i, n < 32,
                                 followed by Gc[i] = (sum_m A^m e_{N-1})_i                  */
    int32_t        levels2_max;   /* largest levels2 in the group                          */
    int32_t        even_b;        /* IIR fast groups: b[1], b[3], ... are exactly zero in every design (every
                                     band-pass from the bilinear transform: numerator k (z^2 - 1)^n) -- the
                                     re-associated kernels of orders 9..16 then issue no multiply-add for them */
    const int32_t *levels1;   /* per design: combine levels of the one-wave kernel (A^(64 2^k)) that matter */
    const int32_t *gtab1_off; /* per design: offset in pows of the one-wave kernel's zero-state table:
                                 G[n][i] = (A^(63-n) B)_i, n < 64, followed by Gc[i]            */
    const int32_t *nb;        /* per design                                               */
    const int32_t *na;        /* per design                                               */
    const int32_t *kind;      /* per design: RTFB_KIND_*                                  */
    const double  *coefs;
    const double  *pows;
};

struct BlockIO {
    double       *out;
    size_t        ld_out;
    const double *in;
    size_t        ld_in;
    size_t        n;
    int           aligned16;  /* in/out/ld allow 16-byte vector access */
    int           runup = 0;  /* iir_scan16_kernel only (RTFilterStream): the rows are consecutive sub-chunks of one
                                 stream; every channel but the first starts from the zero-state response of the
                                 samples in front of its row (kernels.hip): 4 = the 4096 before it, 1 = the last 1024;
                                 + 8 = the first channel too.  0 everywhere else. */
    const double *runup_src = nullptr;   /* with runup == 1: the last 1024 samples in front of row gi (gi >= 1) were
                                            copied to runup_src + (gi - 1) * 1024 beforehand (in-place streams, whose
                                            rows already hold outputs when their successors start) */
};

/* Chained launches of the pipelined kernel (RTFilterBank_updaten_chained): consecutive blocks of a bank are
 * launched on two alternating internal streams, and the workgroups of block k+1 take the slots that block k's
 * workgroups free one by one instead of waiting for the whole launch to drain.  A workgroup of launch k+1 asks
 * launch k's READY QUEUE for a channel whose block-k state is final (ticket = blockIdx.x: entry i is filled by the
 * (i+1)-th workgroup of launch k that finishes), reads that channel's carried state from its CARRY RECORD and
 * leaves the state after block k+1 there.  Every word that crosses between the two launches is self-validating:
 *   queue entry   64 bits: (generation << 32) | (group-local channel + 1)        one 8-byte agent-scope store
 *   carry record  per channel N+2 granules of 16 bytes {value, generation}        one 16-byte sc1 store each
 * so a reader polls until the generation it expects shows up and needs no fence.  Generations grow by one per
 * launch and never repeat during a bank's life. */
struct ChainView {
    const unsigned long long *prev_queue;  /* ready queue of the previous launch; nullptr = chain head: channel =
                                              blockIdx.x, state from the group's plain arrays                       */
    unsigned long long *my_queue;          /* this launch's ready queue (count entries)                            */
    unsigned long long *finished;          /* cumulative count of published channels on my_queue's ring slot       */
    unsigned long long finished_base;      /* its value before this launch                                         */
    unsigned long long *started;           /* cumulative count of this ring slot's workgroups that have started:
                                              the next launch is only enqueued behind a gate kernel that waits
                                              for all of them (no workgroup may spin on one that is not resident) */
    double *records;                       /* count x (N + 2) granules {value, generation}: s[0..N], filtered_value */
    unsigned int gen;                      /* generation this launch writes; it reads gen - 1                      */
    unsigned int *error;                   /* set to nonzero when a bounded wait ran out                           */
};

/* What only the pipelined kernel knows how to use. */
struct PipeExt {
    size_t seg_in, seg_out;                /* doubles between consecutive 2048-sample segments of a channel's block:
                                              2048 for rows; nch * 2048 in the tiled block layout
                                              [segment][channel][2048] (then BlockIO.ld_* = 2048)                  */
    ChainView chain;                       /* chain.records == nullptr: an ordinary launch                         */
    /* Paced launch (tiled layout, one resident round): every workgroup holds the prefetch of its step `it` back
     * until  launch start + it x period  on the chip-wide 100 MHz clock, so that the workgroups of a launch sweep
     * the tiled block in step (bank.hip: pace_period).  Period in 1/1024 ticks; 0 = not paced. */
    unsigned int pace_fx;
};

/* The pipelined kernel for orders 9..16 (pipe16_kernel.hip).  Bank mode: one workgroup per channel, segments in
 * time order, the carry stays in the workgroup.  STREAM mode (exch != nullptr): ONE channel whose 2048-sample
 * segments are dealt round-robin to the grid's workgroups (workgroup w: segments w, w + G, w + 2G, ...), so that
 * at any moment the grid sweeps one contiguous window of the stream; segment m starts from the ZERO-STATE end
 * vector of segment m - 1, which the workgroup that owns m - 1 publishes one step ahead (granules {value,
 * generation}, generation of segment m = gen_base + m + 1) -- exact to 2^-80 for designs whose decay length is at
 * most one segment, the only ones the host sends here.  The grid must be resident at once. */
struct Pipe16Ext {
    double *exch;                          /* [segments of the call][16] granules; nullptr: bank mode              */
    unsigned long long gen_base;
    unsigned int *error;                   /* raised when a bounded wait ran out                                   */
};

/* launchers (kernels.hip).  Each returns the hipError_t of the launch. */
hipError_t launch_iir_seq(int order, const GroupView &g, const BlockIO &io, hipStream_t s);
hipError_t launch_iir_scan(int order, const GroupView &g, const BlockIO &io, hipStream_t s);
hipError_t launch_iir_scan2(int order, const GroupView &g, const BlockIO &io, hipStream_t s);
hipError_t launch_iir_pipe(int order, const GroupView &g, const BlockIO &io, hipStream_t s);
/* the pipelined kernel on whole, 16-byte aligned segments with a segment stride (tiled layout) and / or as one
 * link of a chain; g.count must not exceed pipe_resident_round() for a chained launch */
hipError_t launch_iir_pipe_ext(int order, const GroupView &g, const BlockIO &io, const PipeExt &ext, hipStream_t s);
int pipe_resident_round();
hipError_t launch_chain_gate(const unsigned long long *started, unsigned long long target, unsigned int *error,
                             hipStream_t s);
hipError_t launch_chain_unpack(const double *records, double *state, double *fval, int32_t *init, int count, int N,
                               int state_ld, hipStream_t s);
hipError_t launch_tile_rows(double *dst, const double *src, size_t ld, size_t nch, size_t n, int to_tiles,
                            hipStream_t s);
/* orders 9..16 on whole, 16-byte aligned segments (n a multiple of 2048): bank mode (ext.exch == nullptr) or one
 * stream dealt over `grid` workgroups (<= pipe_resident_round(), <= n / 2048) */
hipError_t launch_iir_pipe16(int order, const GroupView &g, const BlockIO &io, const Pipe16Ext &ext, int grid,
                             hipStream_t s);
hipError_t launch_iir_generic(const GroupView &g, const BlockIO &io, hipStream_t s);
/* fused second-order sections (1 .. kMaxFusedSections per channel, a0 == 1): all sections per sample in one pass */
constexpr int kMaxFusedSections = 4;
hipError_t launch_iir_seq_sos(int sections, const GroupView &g, const BlockIO &io, hipStream_t s);
hipError_t launch_iir_pipe_sos(int sections, const GroupView &g, const BlockIO &io, hipStream_t s);
hipError_t launch_fir(const GroupView &g, const BlockIO &io, hipStream_t s);
/* the matrix-core FIR kernel on the tiled block layout [n / 2048][nch][2048] (io.ld_* = 2048; seg_* = doubles
 * between consecutive tiles of a channel) */
bool fir_tiled_supported(const GroupView &g, size_t n, bool exact);
hipError_t launch_fir_tiled(const GroupView &g, const BlockIO &io, size_t seg_in, size_t seg_out, bool exact,
                            double pace_ns, hipStream_t s);
/* false when launch_fir takes the lane-per-channel kernel, which updates g.state in place (no ping-pong swap) */
bool fir_uses_state_swap(const GroupView &g, const BlockIO &io);
hipError_t launch_reset(const GroupView &g, hipStream_t s);
hipError_t launch_synth(double *dst, size_t ld, size_t nch, size_t n, uint64_t seed,
                        uint64_t chan0, uint64_t n0, hipStream_t s);

hipError_t launch_copy_rows(double *out, size_t ld_out, const double *in, size_t ld_in, size_t nch,
                            size_t n, int threads, int seg, size_t seg_stride, hipStream_t s);
hipError_t launch_state_pack(double *pack, const double *state, const double *fval, const int32_t *init,
                             int ld, hipStream_t s);
hipError_t launch_state_unpack(const double *pack, double *state, double *fval, int32_t *init, int ld,
                               hipStream_t s);
hipError_t launch_compose_states(double *dst, size_t dst_ld, const double *src0, const double *v, size_t v_ld,
                                 const double *M, int N, int count, int write_all, hipStream_t s);
hipError_t launch_split_zero(double *state, int32_t *init, size_t count, int ld, hipStream_t s);
hipError_t launch_split_seed(double *state, double *fval, int32_t *init, const double *one_state, const double *one_fval,
                             const int32_t *one_init, const double *warm, size_t count, int ld, int with_warm,
                             int keep_init, hipStream_t s);
hipError_t launch_split_finish(double *one_state, double *one_fval, int32_t *one_init, const double *state,
                               const double *fval, size_t count, int ld, hipStream_t s);
hipError_t launch_transpose(double *dst, size_t ld_dst, const double *src, size_t ld_src, size_t rows, size_t cols,
                            hipStream_t s);
bool iir_scan_supported(int order);

}  // namespace dsp

#endif
