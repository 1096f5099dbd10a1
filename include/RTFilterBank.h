/*
 * RTFilterBank.h -- additive MI355X (gfx950) API of libdsp: many independent
 * RTFilter channels advanced together on device-resident sample blocks.
 *
 * The reference has no multichannel entry point; a caller loops
 *     RTFilter_updaten(out_c, filter_c, in_c, n)      (ref: include/RTFilter.h:89,
 *                                                       src/RTFilter.c:111-119)
 * once per filter.  RTFilterBank_updaten is defined as exactly that loop over
 * the bank's channels -- same outputs, same end state, same `filtered_value`
 * and `initialized` per channel -- executed by hand-written HIP kernels, with
 * the filter state living in HBM between calls so that consecutive blocks
 * stream (the "block-streaming state manager" of the design).
 *
 * Everything here is plain C ABI: opaque handle, raw device pointers, sizes,
 * int status.  No torch / C++ types cross this boundary.
 *
 * Status codes extend the reference's convention (src/RTFilter.c:112-114,
 * 1217-1223: 0 ok, 1 bad argument, 2 bank/size problem).
 */
#ifndef RTFILTERBANK_H
#define RTFILTERBANK_H

#include <stddef.h>
#include <stdint.h>
#include "RTFilter.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RTFB_OK            0
#define RTFB_EINVAL        1   /* NULL / malformed argument (reference code 1)              */
#define RTFB_ESIZE         2   /* a size the bank cannot take (reference code 2)            */
#define RTFB_EUNSUPPORTED  3   /* filter carries callbacks that are not the library's own   */
#define RTFB_EDEVICE       4   /* HIP runtime error or no gfx950 device; see last_error     */
#define RTFB_ENOMEM        5

typedef struct RTFilterBank RTFilterBank;

/* What a channel computes: mirrors the (update, initialize) callback pair a
 * reference filter carries. */
#define RTFB_KIND_IIR          0   /* RTIIRFilter_update   + RTIIRFilter_stable_init  */
#define RTFB_KIND_FIR          1   /* RTFIRFilter_update   + RTFIRFilter_stable_init  */
#define RTFB_KIND_FIR_PARTIAL  2   /* RTFIRFilter_update   + RTFIRFilter_partial_init */
#define RTFB_KIND_SOS          3   /* a cascade of second-order sections (FILTER_SOS; RTFilterDesign.h): nb = na =
                                      3 x sections, b = the sections' numerators one after the other, a likewise
                                      their denominators.  Every section is executed as an RTIIRFilter with
                                      nb = na = 3 (RTIIRFilter_update + RTIIRFilter_stable_init) on the output of
                                      the section before it; state: 3 slots per section, filtered_value of the
                                      last section.  Cascades of up to 4 sections with a0 == 1 in every section
                                      are FUSED: all sections per sample, the block read and written once (the
                                      cascade as one state-space system of 2 states per section on the pipelined
                                      kernel, or the bit-exact lane-per-channel kernel); longer cascades run one
                                      pass over the block per section.  The third state slot of a section (read,
                                      never written; 0 in every state the library produces) is honoured by the
                                      bit-exact kernels only. */

/* One distinct coefficient set; channels reference designs by index. */
typedef struct RTFilterDesign {
    int           kind;   /* RTFB_KIND_*                                   */
    size_t        nb;     /* taps in b (FIR: the filter length)            */
    size_t        na;     /* taps in a (IIR only; 0 for FIR)               */
    const double *b;      /* host pointer, nb doubles                      */
    const double *a;      /* host pointer, na doubles (IIR)                */
} RTFilterDesign;

/* Kernel choice.
 * AUTO        fastest kernels that meet the 1e-12 relative-to-peak bar: IIR channels on the
 *             time-parallel scan kernels when their group has fewer than 131072 channels (orders above
 *             8: at any width once a call brings >= 4096 samples, else below 65536 channels) and the
 *             design passes the conditioning probe,
 *             else on the bit-exact sequential kernel; FIR channels of >= 2 taps on the
 *             Toeplitz-tile FP64 MFMA kernel (4.9 TB/s at any short length, where the bit-exact kernel
 *             falls from 4.7 at 2 taps to 3.5 at 32); short blocks on wide banks (>= 16384 channels,
 *             <= 4096 samples, <= 32 taps) stay on the bit-exact lane-per-channel kernel in every mode.
 * SEQUENTIAL  bit-exact kernels only (reference operation order) for every channel (FIR of <= 128 taps on banks
 *             of >= 512 channels and whole 4096-sample pieces: the streaming kernel, 4.4 TB/s at 32 taps, 2.8 at
 *             64, 1.5 at 128, where the unfused arithmetic is the bound).
 * SCAN        like AUTO but IIR channels always on the scan kernels: the conditioning probe is bypassed, and with
 *             it every accuracy promise -- the re-associated arithmetic then sits at the design's own conditioning
 *             floor, which can be far above 1e-12 (19 of 200 random Butterworth designs of orders 10..16 exceeded
 *             it in this mode, none under AUTO: DESIGN.md section 5; the demo's narrow band-pass is 5e-8 off).
 *             For callers who have validated their designs themselves. */
#define RTFB_MODE_AUTO        0
#define RTFB_MODE_SEQUENTIAL  1
#define RTFB_MODE_SCAN        2

/* Build a bank from host filters.  Each entry must be an RTIIRFilter or
 * RTFIRFilter whose callbacks are this library's own (pointer identity, the
 * "acceleration rule"); anything else -> RTFB_EUNSUPPORTED.  Coefficients and
 * the CURRENT state / filtered_value / initialized are copied to `device`.
 * The filter pointers are retained for RTFilterBank_sync_to_host and must
 * outlive the bank. */
EXPORT int RTFilterBank_from_filters(RTFilterBank **bank, RTFilter *const *filters,
                                     size_t nch, int device);

/* Build a bank from packed tables: design_of_channel[c] in [0, ndesign).
 * All channels start in the state RT{IIR,FIR}Filter_init leaves (state zero,
 * initialized = FILTER_RESET). */
EXPORT int RTFilterBank_from_designs(RTFilterBank **bank, const RTFilterDesign *designs,
                                     size_t ndesign, const int32_t *design_of_channel,
                                     size_t nch, int device);

/* Conditioning probe used by RTFB_MODE_AUTO (host arithmetic, no device needed, produces no filter output):
 * runs the re-associated algorithm of the scan kernel the design would execute on -- same row length, rows per
 * segment, tables, order of the fused multiply-adds and skipped combine levels -- against the plain recurrence on
 * pseudo-random input of unit scale and reports max|dy|/max|y|.  Designs above 1e-13 are kept on the sequential
 * kernel in AUTO mode.  dsp_scan_probe picks the kernel of a bank with few designs (pipelined kernel for orders
 * <= 8, matrix-core scan for 9..16); dsp_scan_probe_kernel names it.  RTFB_EUNSUPPORTED unless IIR with
 * na == nb <= 17.
 * What AUTO promises: 1e-12 relative to the larger of the output peak and the scale of the filter's state.  For
 * zero-mean broadband input those coincide; a stream dominated by DC or out-of-band content that the design
 * rejects keeps the state at the input's scale while the output is small, and the re-association error stays
 * relative to the former.  RTFB_MODE_SEQUENTIAL is the bit-exact mode. */
#define RTFB_SCAN_KERNEL_PIPE    0   /* orders <= 8, banks of <= 1024 distinct designs */
#define RTFB_SCAN_KERNEL_SCAN2   1   /* orders <= 8, banks of more designs             */
#define RTFB_SCAN_KERNEL_SCAN16  2   /* orders 9..16                                   */
EXPORT int dsp_scan_probe(const RTFilterDesign *design, double *rel_error);
EXPORT int dsp_scan_probe_kernel(const RTFilterDesign *design, int kernel, double *rel_error);

/* Host arithmetic only: doubles of per-design scan tables a bank of these designs carries at most (fast IIR
 * designs: matrix powers, zero-state response tables, row propagators).  The kernels address those tables with
 * 32-bit offsets; RTFB_ESIZE (here and from the bank constructors) when a bank would exceed 2^31 doubles --
 * about 1.3e6 distinct order-8 or 4.4e5 distinct order-16 designs.  Groups of >= 131072 channels of order <= 8
 * run the lane-per-channel kernel in every mode and carry no tables at all. */
EXPORT int dsp_bank_table_doubles(const RTFilterDesign *designs, size_t ndesign, size_t *doubles);

EXPORT void   RTFilterBank_destroy(RTFilterBank *bank);
EXPORT size_t RTFilterBank_channels(const RTFilterBank *bank);
EXPORT int    RTFilterBank_set_mode(RTFilterBank *bank, int mode);

/* For every channel c: RTFilter_updaten(out + c*ld_out, filter_c, in + c*ld_in, n).
 * in/out are DEVICE pointers on the bank's device, channel-major, ld_* in
 * doubles (>= n); in == out (same ld) is allowed as in the reference.
 * `hip_stream` is a hipStream_t (NULL = default stream); the call enqueues
 * and returns.  n == 0 is a no-op. */
EXPORT int RTFilterBank_updaten(RTFilterBank *bank, double *out, size_t ld_out,
                                const double *in, size_t ld_in, size_t n, void *hip_stream);

/* RTFilterBank_updaten for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  (No counterpart in the reference.)  Same arguments, same results,
 * same state carried; this entry point calls no hipMalloc / hipFree / pool routine.  The one place where
 * RTFilterBank_updaten may allocate -- an in-place call (in and out overlap) on FIR channels, whose kernels read
 * from a private copy of the input because FIR outputs are produced out of time order -- stages that copy in a
 * static device array of the library instead: channels x n <= RTFilterBank_noalloc_max_doubles() (4 Mi doubles =
 * 32 MiB, held by every GPU the library runs on until the process ends), else RTFB_ESIZE.  Every call on a GPU shares that
 * GPU's array: calls that overlap in time on one GPU (different streams or threads) race on it; calls on one
 * stream are ordered.  Calls that are not in place, or touch no FIR channel, need no staging and have no size
 * limit.  A captured call may be REPLAYED any number of times: every call, IIR or FIR, leaves the carried state in
 * the one buffer the next call reads (the FIR end-state kernels build it in a second buffer and commit it at the
 * end of the call), so each replay continues the stream by one block. */
EXPORT int RTFilterBank_updaten_noalloc(RTFilterBank *bank, double *out, size_t ld_out,
                                        const double *in, size_t ld_in, size_t n, void *hip_stream);
EXPORT size_t RTFilterBank_noalloc_max_doubles(void);

/* ---- tiled block layout, chained block updates (no counterpart in the reference) ---------------------------
 * Both serve the streaming form of RTFilter_updaten (ref: src/RTFilter.c:111-119 called block after block on the
 * same filters) on banks that the pipelined IIR kernel runs: ONE launch group of IIR channels of order <= 8
 * (na == nb), at most 1024 distinct designs, admitted by the conditioning probe (or RTFB_MODE_SCAN).  Results are
 * the bits RTFilterBank_updaten produces for the same blocks.
 *
 * TILED LAYOUT.  A block of n samples (n a multiple of RTFB_TILE) of all nch channels is stored
 * [n / RTFB_TILE][nch][RTFB_TILE]: sample k of channel c at ((k / RTFB_TILE) * nch + c) * RTFB_TILE + k % RTFB_TILE.
 * The workgroups that run together then sweep one contiguous window of HBM per step instead of nch streams a row
 * pitch apart (measured on the copy twin: 5.9 vs 5.45 TB/s at 1024 x 64K, profiles/r3_tiled_copy.txt).  in == out
 * is allowed.  dsp_tile_rows / dsp_untile_rows convert between rows (pitch ld) and tiles on the device.
 * Banks wider than the workgroups the GPU holds at once (1024 on MI355X) are launched round by round, 1024
 * channels each: 65 536 x 64K runs at 5.56 TB/s where dense rows give 5.24 (profiles/r3_config4_tiled.txt).
 * _updaten_tiled (not the chained entry) also serves FIR banks: ONE launch group of FIR designs, n a multiple of
 * 2 RTFB_TILE, in == out or disjoint.  Up to 32 taps (either initialiser) they run on the streaming FIR
 * kernel (one workgroup per channel walking its tiles, round by round on the schedule like the IIR banks):
 * 65 536 x 64K at 32 taps 5.55 TB/s against 5.0 on dense rows, 7 taps 5.95 (profiles/r3_fir_tile.txt).  Its
 * arithmetic is the reference's -- bit-exact -- in RTFB_MODE_SEQUENTIAL and, in the other modes, up to 16 taps
 * (where that costs nothing); 17..32 taps use fused multiply-adds outside RTFB_MODE_SEQUENTIAL (1e-12 class).
 * 33..128 taps: the same kernel where the matrix cores may not run -- RTFB_MODE_SEQUENTIAL (bit-exact) and banks
 * with a partial_init design (fused outside RTFB_MODE_SEQUENTIAL).  33..2047 taps otherwise (stable_init designs,
 * not RTFB_MODE_SEQUENTIAL): the matrix-core kernel walks the tiles.  Above 128 taps RTFB_MODE_SEQUENTIAL and
 * partial_init banks get RTFB_EUNSUPPORTED.
 * Other banks: RTFB_EUNSUPPORTED; a block length that does not fit: RTFB_ESIZE. */
#define RTFB_TILE 2048
EXPORT int RTFilterBank_updaten_tiled(RTFilterBank *bank, double *out, const double *in, size_t n, void *hip_stream);
/* The tiled layout pays while the workgroups of a launch stay in step, so tiled launches of banks that fill the GPU
 * (512 .. 1024 channels on MI355X) run on a schedule: no workgroup requests segment k + 2 of its channel earlier
 * than launch start + k x period on the chip-wide 100 MHz clock.  A performance hint only -- results do not depend
 * on it.  The library picks the period (the time one round of segments takes at the lock-step copy rate);
 * ns_per_segment > 0 sets it, 0 switches pacing off, < 0 returns to the library's choice. */
EXPORT int RTFilterBank_set_pace(RTFilterBank *bank, double ns_per_segment);
EXPORT int dsp_tile_rows(double *tiles, const double *rows, size_t ld, size_t nch, size_t n, void *hip_stream);
EXPORT int dsp_untile_rows(double *rows, size_t ld, const double *tiles, size_t nch, size_t n, void *hip_stream);

/* CHAINED UPDATES.  RTFilterBank_updaten launches block k+1 behind block k in stream order, so every launch fills
 * and drains the GPU alone (8 % of a 64K-sample step of a 1024-channel bank).  _updaten_chained enqueues the
 * blocks of a stream on two internal streams of the bank instead: the workgroups of block k+1 take the slots that
 * block k's workgroups free one by one, each picking a channel whose block-k state is final (a ready queue and
 * per-channel carry records in HBM; dsp_internal.h: ChainView).  Same arguments as _updaten plus
 *   flags        RTFB_CHAIN_TILED: in / out are in the tiled layout above (ld_* ignored)
 *   done_event   if not NULL: receives a hipEvent_t (owned by the bank, valid until four more chained calls)
 *                that completes when this block's outputs are written.
 * Ordering: the call is ordered behind everything enqueued on `hip_stream` before it (the producer of `in`);
 * nothing enqueued on `hip_stream` afterwards waits for it -- consumers wait on done_event, or call
 * RTFilterBank_chain_join(bank, stream), which makes `stream` wait for all chained calls so far.  Every other
 * entry point of the bank (updaten, reset, get/set_state, sync_to_host, destroy) joins first on the stream it is
 * given.  n must be a multiple of RTFB_TILE, rows 16-byte aligned, and the bank no wider than the workgroups the
 * GPU holds at once (1024 on MI355X); a call that does not qualify runs in stream order on `hip_stream` with the
 * same results (done_event still valid).  One thread drives a bank's chain.  All waits between launches are
 * bounded (2 s): if one runs out the bank reports RTFB_EDEVICE from the next call and must be rebuilt. */
#define RTFB_CHAIN_TILED 1u
EXPORT int RTFilterBank_updaten_chained(RTFilterBank *bank, double *out, size_t ld_out, const double *in,
                                        size_t ld_in, size_t n, unsigned flags, void *hip_stream,
                                        void **done_event);
EXPORT int RTFilterBank_chain_join(RTFilterBank *bank, void *hip_stream);

/* The same loop for HOST arrays -- what an unchanged caller of the reference holds: in/out are ordinary
 * (pageable) host memory, channel-major, ld_* in doubles, in == out allowed.  The block is moved through the GPU in
 * chunks of time (all channels x nt samples each, so the carried state streams from chunk to chunk): the CPU
 * stages each chunk into a ring of two pinned buffers while the GPU works on the previous one, H2D / filter / D2H
 * run on three streams, and the results leave through a second pinned ring.  Bounded by the host link and the
 * host's memcpy rate (PCIe: 16 B/sample), never by the kernels; blocks until out[] is complete.  The rings
 * (4 x 64 MiB pinned, 4 x 64 MiB on the device) are created by the first call.  (ref: src/RTFilter.c:111-119.) */
EXPORT int RTFilterBank_updaten_host(RTFilterBank *bank, double *out, size_t ld_out,
                                     const double *in, size_t ld_in, size_t n);

/* ---- per-sample producers ------------------------------------------------------------------
 * The reference's real-time entry is RTFilter_update(filter, sample): one call per filter and sample
 * (src/RTFilter.c:98-108; python/dsp.py:6-13,35-38 wraps it in an iterator).  A GPU cannot serve that call by
 * call; RTFilterBankFeed coalesces it: the producer pushes FRAMES (frame t = one sample for every channel of the
 * bank, nch doubles, time-major), the feed collects `block_frames` of them in pinned memory, sends the block
 * through the device (H2D, frames -> channel rows, RTFilterBank_updaten with the carried state, rows -> frames,
 * D2H) and the consumer pops filtered frames in the same order.  Outputs equal RTFilter_update applied per
 * channel and frame; latency is one block.  Two blocks may be in flight: when a third would be needed
 * before the oldest was popped, _push stops there and reports the frames it took in *accepted (RTFB_OK; with
 * accepted == NULL that situation is RTFB_ESIZE).  _pop never blocks (returns what is ready), _wait blocks until the
 * oldest submitted block is back; _flush submits a partial block and waits for everything in flight.  One producer/consumer thread per feed.
 * Destroy a feed before its bank.  RTFilterBank_updaten_host and the feed run the bank on streams of their own:
 * updates that the caller has enqueued on other streams for the same bank must have completed before (the carried
 * state is not ordered between streams). */
typedef struct RTFilterBankFeed RTFilterBankFeed;
EXPORT int    RTFilterBankFeed_create(RTFilterBankFeed **feed, RTFilterBank *bank, size_t block_frames);
EXPORT void   RTFilterBankFeed_destroy(RTFilterBankFeed *feed);
EXPORT int    RTFilterBankFeed_push(RTFilterBankFeed *feed, const double *frames, size_t nframes, size_t *accepted);
EXPORT int    RTFilterBankFeed_pop(RTFilterBankFeed *feed, double *frames, size_t max_frames, size_t *got);
EXPORT size_t RTFilterBankFeed_available(RTFilterBankFeed *feed);    /* filtered frames ready to pop */
EXPORT size_t RTFilterBankFeed_pending(RTFilterBankFeed *feed);      /* submitted blocks not yet fully popped: 0..2 */
EXPORT int    RTFilterBankFeed_wait(RTFilterBankFeed *feed);      /* until the oldest submitted block is back */
EXPORT int    RTFilterBankFeed_flush(RTFilterBankFeed *feed);

/* RTFilter_reset on every channel (state untouched, src/RTFilter.c:55-58). */
EXPORT int RTFilterBank_reset(RTFilterBank *bank, void *hip_stream);

/* Copy device state into the host structs given to _from_filters so that they
 * hold what the reference would have left (state[], filtered_value,
 * initialized).  Synchronises the stream. */
EXPORT int RTFilterBank_sync_to_host(RTFilterBank *bank, void *hip_stream);

/* Raw state access (host buffers): state is nch x ld doubles (row c = channel c,
 * first max(na,nb) / nb entries meaningful), fval nch doubles, init nch ints. */
EXPORT size_t RTFilterBank_state_ld(const RTFilterBank *bank);
EXPORT int RTFilterBank_get_state(RTFilterBank *bank, double *state, double *fval, int32_t *init,
                                  void *hip_stream);
EXPORT int RTFilterBank_set_state(RTFilterBank *bank, const double *state, const double *fval,
                                  const int32_t *init, void *hip_stream);

/* Thread-local description of the last failure in this API. */
EXPORT const char *RTFilterBank_last_error(void);

/* ---- device memory pool --------------------------------------------------
 * Size-bucketed caching allocator over hipMalloc for sample blocks and bank
 * tables (sized for 288 GB of HBM3E: blocks are reused, not re-allocated per
 * call).  Thread-safe. */
EXPORT int    dsp_pool_alloc(void **dev_ptr, size_t bytes, int device);
/* dsp_pool_free: like hipFree, synchronises the device first (no stream can still be using the block when it is
 * handed out again).  dsp_pool_free_async: stream-ordered and non-blocking -- the block returns to the cache at
 * once but is passed over by dsp_pool_alloc until the work enqueued on `hip_stream` before this call has
 * completed (an event recorded there gates the reuse). */
EXPORT int    dsp_pool_free(void *dev_ptr);
EXPORT int    dsp_pool_free_async(void *dev_ptr, void *hip_stream);
EXPORT int    dsp_pool_trim(int device);               /* release cached blocks to the driver */
EXPORT size_t dsp_pool_bytes_in_use(int device);
EXPORT size_t dsp_pool_bytes_cached(int device);

/* ---- helpers for harnesses ------------------------------------------------
 * Deterministic synthetic input:
 *   x[c][k] = (double)(int64)(splitmix64(seed ^ ((chan0+c) << 32) + (n0+k)) >> 11) * 2^-52 - 1
 * written on device to dst + c*ld + k, c < nch, k < n. */
EXPORT int dsp_synth_fill(double *dst, size_t ld, size_t nch, size_t n,
                          uint64_t seed, uint64_t chan0, uint64_t n0, void *hip_stream);
/* Compute-free twin of the IIR scan kernels' memory access pattern: one workgroup per row walks
 * its n doubles in `seg`-double pieces (two pieces of loads in flight), out[c*ld_out+k] =
 * in[c*ld_in+k].  bench.py times it to report the streaming-copy ceiling of the device next to the
 * roofline fraction.  threads in {64,128,256}; seg divides n; rows 16-byte aligned. */
EXPORT int dsp_copy_rows(double *out, size_t ld_out, const double *in, size_t ld_in, size_t nch,
                         size_t n, int threads, int seg, void *hip_stream);
/* dsp_copy_rows for the tiled block layout [n / RTFB_TILE][nch][RTFB_TILE] (256 threads, one tile per piece). */
EXPORT int dsp_copy_tiles(double *out, const double *in, size_t nch, size_t n, void *hip_stream);
EXPORT int dsp_device_count(void);
EXPORT int dsp_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, int device);
EXPORT int dsp_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, int device);
EXPORT int dsp_stream_synchronize(void *hip_stream, int device);

#ifdef __cplusplus
}
#endif

#endif /* RTFILTERBANK_H */
