/*
 * RTFilterStream.h -- additive MI355X (gfx950) API of libdsp: ONE RTFilter advanced over a
 * very long device-resident sample stream, time-parallel.
 *
 * Replaces the reference call
 *     RTFilter_updaten(out, filter, in, n)              (ref: include/RTFilter.h:89,
 *                                                         src/RTFilter.c:111-119)
 * for one filter and a huge n (BASELINE.json config 5: 2^30 samples, 16th-order IIR), which the
 * reference can only run sample after sample on one core.
 *
 * How: the stream is cut into C consecutive sub-chunks that are filtered as the C channels of an
 * RTFilterBank (row pitch = sub-chunk length, i.e. the stream itself, no copy).  The start state of
 * sub-chunk c is the zero-state response of the M samples in front of it, where M is the design's
 * decay length: the smallest power of two with max|A^M| < 2^-80 for the homogeneous recurrence
 * s' = A s of RTIIRFilter_update (ref: src/RTFilter.c:417-433).  What the truncation drops is
 * below 2^-80 of the state, far inside the 1e-12 bar.  (Up to 1024 sub-chunks, the run-ups in a launch of their
 * own; orders 9..16: sub-chunks of 2^17 samples at most whose run-ups the scan kernel computes itself from the 1024
 * or 4096 samples in front of each -- one launch; in place the former are set aside first.)  Designs that decay too slowly for that (or not within
 * 2^20 samples) have their chunk maps composed exactly instead: a first pass from the zero state gives every
 * sub-chunk's affine map s_end = A^Lc s_start + v, the start states follow by composition, a second pass filters
 * (twice the traffic).  Designs that fail the scan conditioning probe, in-place calls without decay and calls too
 * short to split run as a one-channel RTFilterBank (correct, one workgroup).  FIR designs need none of this
 * (their bank kernels already split time) and go straight to a one-channel bank.
 *
 * Across GPUs (one process per GPU, rank r owns samples [r L, (r+1) L)): every rank but the last
 * computes RTFilterStream_tail_state of its own chunk -- the state hand-off vector, N+3 doubles --
 * and sends it to rank r+1 (RCCL/xGMI send-recv or all_gather), which imports it and runs
 * RTFilterStream_updaten on its chunk.  All ranks then filter concurrently.
 *
 * Plain C ABI: opaque handle, raw device pointers, sizes, int status (RTFB_* of RTFilterBank.h).
 */
#ifndef RTFILTERSTREAM_H
#define RTFILTERSTREAM_H

#include <stddef.h>
#include <stdint.h>
#include "RTFilterBank.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RTFilterStream RTFilterStream;

/* A stream filter in the state RT{IIR,FIR}Filter_init leaves (state zero, initialized = FILTER_RESET). */
EXPORT int RTFilterStream_from_design(RTFilterStream **stream, const RTFilterDesign *design, int device);
/* From a host filter (library callbacks only, else RTFB_EUNSUPPORTED): coefficients and the CURRENT
 * state / filtered_value / initialized are copied; the pointer is kept for _sync_to_host. */
EXPORT int RTFilterStream_from_filter(RTFilterStream **stream, RTFilter *filter, int device);
EXPORT void RTFilterStream_destroy(RTFilterStream *stream);

/* == RTFilter_updaten(out, filter, in, n) with DEVICE pointers; in == out allowed; enqueues on
 * `hip_stream` (a hipStream_t, NULL = default) and returns.  State carries to the next call. */
EXPORT int RTFilterStream_updaten(RTFilterStream *stream, double *out, const double *in, size_t n,
                                  void *hip_stream);
/* The same for HOST buffers, i.e. the reference call as an unchanged caller makes it:
 * in/out are ordinary (pageable) host arrays of n doubles, in == out allowed.  The stream is moved through
 * the GPU in chunks of 2^23 samples, double-buffered (copy-in of chunk k+1 and copy-out of chunk k-1 overlap
 * the filtering of chunk k); throughput is bounded by the host link (PCIe: ~16 B/sample at a few tens of GB/s),
 * never by the kernels.  Blocks until out[] is complete. */
EXPORT int RTFilterStream_updaten_host(RTFilterStream *stream, double *out, const double *in, size_t n);
/* RTFilter_reset (src/RTFilter.c:55-58): initialized = FILTER_RESET, state untouched. */
EXPORT int RTFilterStream_reset(RTFilterStream *stream, void *hip_stream);
/* Host structs given to _from_filter take state, filtered_value, initialized.  Synchronises. */
EXPORT int RTFilterStream_sync_to_host(RTFilterStream *stream, void *hip_stream);

/* Decay length M in samples; 0 = the design does not qualify for the time split. */
EXPORT size_t RTFilterStream_decay_length(const RTFilterStream *stream);
/* Sub-chunks the last _updaten call used for its largest piece (1 = sequential path); diagnostics. */
EXPORT size_t RTFilterStream_last_split(const RTFilterStream *stream);

/* ---- state hand-off (device side) --------------------------------------------------------
 * A hand-off vector is RTFilterStream_pack_len() doubles in device memory:
 *   state[0 .. state_ld), filtered_value, (double)initialized. */
EXPORT size_t RTFilterStream_pack_len(const RTFilterStream *stream);
EXPORT int RTFilterStream_export_state(RTFilterStream *stream, double *pack_dev, void *hip_stream);
EXPORT int RTFilterStream_import_state(RTFilterStream *stream, const double *pack_dev, void *hip_stream);
/* The hand-off vector a filter would hold after in[0..n) to within 2^-80 of its state, computed
 * from the last min(n, M) samples alone (zero-state response); does not touch the stream's own
 * state.  RTFB_EUNSUPPORTED when the design does not decay (decay_length == 0) or n < M: the
 * caller must then chain the ranks one after another with export/import of the true end state. */
EXPORT int RTFilterStream_tail_state(RTFilterStream *stream, const double *in, size_t n,
                                     double *pack_dev, void *hip_stream);

/* ---- exact composition of chunk maps (no decay assumption) ----------------------
 * A chunk of L samples maps the state affinely: s_end = A^L s_start + v, v = the chunk's end state when it
 * starts from zero.  RTFilterStream_updaten uses this across its sub-chunks when the design does not decay
 * fast enough for the shortcut above (twice the traffic, still C-fold parallel).  Across ranks:
 *   rank 0         runs its chunk from the stream's state; its hand-off vector is its true end state;
 *   rank r >= 1    _zero_state, _updaten on its chunk (outputs to be overwritten), _export_state = v_r;
 *   all_gather of the vectors;
 *   rank r >= 1    _compose_state(packs, stride, r, L): state = A^L( ... A^L(A^L p_0 + p_1) + p_2 ... ) + p_(r-1)
 *                  for the r vectors p_0 .. p_(r-1) of the ranks before it (chunks 1 .. r-1 of L samples each),
 *                  then _updaten on its chunk again, now from the true start state.
 * _exact_split: 1 if the design qualifies (IIR, na == nb <= 17, passes the conditioning probe). */
EXPORT int RTFilterStream_exact_split(const RTFilterStream *stream);
EXPORT int RTFilterStream_zero_state(RTFilterStream *stream, void *hip_stream);
EXPORT int RTFilterStream_compose_state(RTFilterStream *stream, const double *packs_dev, size_t stride,
                                        size_t count, size_t chunk_len, void *hip_stream);

/* ---- the time split across GPUs as ONE call (no counterpart in the reference) ---------------------------------
 * The reference call served is RTFilter_updaten on one filter (include/RTFilter.h:89) over a stream whose
 * consecutive chunks live on consecutive ranks: rank r holds lens[r] samples, the chunks in rank order are the
 * stream.  _split_updaten runs the protocol above on the caller's stream with the caller's collective -- a C
 * program needs no Python for it: every rank calls it with the same lens / world and its own rank, in / out = its
 * chunk (in == out allowed).  The hand-off mode follows from the design and the partition alone (no negotiation):
 *   decay tail      the design decays (decay_length > 0) and every rank but the last holds >= decay_length samples:
 *                   ONE all-gather of 2 x pack_len doubles per rank, all ranks filter concurrently;
 *   composed maps   otherwise, when _exact_split and the ranks 1 .. world-2 hold equal lengths: one all-gather between
 *                   two passes on ranks >= 1;
 *   else            RTFB_EUNSUPPORTED: chain the ranks with _export_state / _import_state.
 * The stream's end state is left on the LAST rank and rides back to rank 0 inside the next call's all-gather, so
 * consecutive calls continue one stream; _split_end_state puts it on every rank (before a different partition, or
 * before _sync_to_host / single-GPU calls).
 * allgather(ctx, send_dev, recv_dev, count, hip_stream): gather `count` doubles from every rank's send_dev into
 * recv_dev (rank-major, world x count doubles, device memory owned by the stream), ordered on hip_stream (or
 * complete on return); 0 = success.  Buffers are allocated on the first call per world size; later calls enqueue only.
 * _rccl: the same with ncclAllGather on an ncclComm_t (RCCL is looked up at run time, libdsp.so does not link it). */
typedef int (*RTFilterStream_allgather_fn)(void *ctx, const double *send_dev, double *recv_dev, size_t count,
                                           void *hip_stream);
EXPORT int RTFilterStream_split_updaten(RTFilterStream *stream, double *out, const double *in, const size_t *lens,
                                        int rank, int world, RTFilterStream_allgather_fn allgather, void *ctx,
                                        void *hip_stream);
EXPORT int RTFilterStream_split_end_state(RTFilterStream *stream, int rank, int world,
                                          RTFilterStream_allgather_fn allgather, void *ctx, void *hip_stream);
EXPORT int RTFilterStream_split_updaten_rccl(RTFilterStream *stream, double *out, const double *in, const size_t *lens,
                                             int rank, int world, void *nccl_comm, void *hip_stream);
EXPORT int RTFilterStream_split_end_state_rccl(RTFilterStream *stream, int rank, int world, void *nccl_comm,
                                               void *hip_stream);

/* ---- interleaved segments (orders 9..16 whose decay length is at most 2048 samples; OFF by default) ----------
 * An alternative to the contiguous time split for such designs: RTFilterStream_updaten deals the stream's
 * 2048-sample segments round-robin over `workgroups` workgroups of the two-wave kernel of pipe16_kernel.hip, so
 * that at any moment the grid works on ONE contiguous window of the stream, and each segment takes its start state
 * from the workgroup that owns the segment in front of it (that segment's zero-state response, exact to 2^-80 like
 * the run-up it replaces; no run-up launch).  Same results to 1e-12 (tests/test_gpu_pipe16.py); measured on MI355X it
 * is NOT faster than the default, which is why it is opt-in.  The workgroups of such a launch wait for
 * each other and must be resident at once: at most 4 per CU (clamped), and not beside another kernel that fills
 * the GPU -- a wait that runs out after 2 s makes this and every later call on the stream return RTFB_EDEVICE.
 * Device memory: 256 bytes per segment of the longest call, kept with the stream.
 * _set_interleave: workgroups > 0 turns the mode on, <= 0 off.  _interleaved: 1 if the design qualifies. */
EXPORT int RTFilterStream_set_interleave(RTFilterStream *stream, int workgroups);
EXPORT int RTFilterStream_interleaved(const RTFilterStream *stream);

/* Host access to the carried state (state_ld doubles, see RTFilterStream_pack_len() - 2). */
EXPORT int RTFilterStream_get_state(RTFilterStream *stream, double *state, double *fval, int32_t *init,
                                    void *hip_stream);
EXPORT int RTFilterStream_set_state(RTFilterStream *stream, const double *state, const double *fval,
                                    const int32_t *init, void *hip_stream);

#ifdef __cplusplus
}
#endif

#endif /* RTFILTERSTREAM_H */
