/*
 * tmpnn_debug.h — measurement and experiment hooks of libtmpnn.so. NOT part of the operator boundary (include/tmpnn.h):
 * nothing in the product path calls these; bench.py (per-kernel HIP-event timing for the roofline leg) and the scripts
 * under tools/ do.
 */
#ifndef TMPNN_DEBUG_H
#define TMPNN_DEBUG_H

#include "tmpnn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-kernel timing ----------------------------------------------------------------------------
 * Optional per-kernel timing with HIP events recorded on the launch stream around every kernel the
 * library launches (bench.py's roofline leg; not thread-safe; off by default). enable(1) starts a
 * fresh recording, enable(0) stops. fetch() waits for the recorded events, aggregates by kernel name
 * into the caller's arrays (up to `capacity` rows, names are static strings), clears the recording and
 * returns the number of rows (or a negative error). */
int tmpnn_profile_enable(int on);
int tmpnn_profile_fetch(const char **names, double *total_ms, int64_t *launches, int capacity);
/* Restrict the recording to ONE kernel name (as fetch() reports it; NULL or "" = every kernel again). Two event records
 * per launch cost the stream ~2 us each: 40 of them in a 20-launch forward are 2.7 % of bench.py's step, 6 around the
 * three launches of the dominant kernel 0.4 %. */
int tmpnn_profile_select(const char *name);

/* GEMM core probe: Y[t] = reps x (X[t] W^T) for T tiles of [48,128] and one [128,128] weight; mode 0 = exact fp32 MFMA,
 * mode 1 = six-term bf16x3 split MFMA, mode 2 = three-term f16x2 split MFMA (tmpnn_split.h). For accuracy / speed
 * comparisons of the matrix-core paths. */
int tmpnn_gemm_probe(int mode, const float *X, const float *W, float *Y, int64_t T, int reps, tmpnn_stream_t stream);

/* Effective shader clock under a saturated fp32-MFMA stream (192 v_mfma_f32_16x16x4_f32 per iteration per wavefront,
 * 4 wavefronts per workgroup): out[2b] = shader cycles of workgroup b, out[2b+1] = the same interval in 100 MHz ticks.
 * `sink` (>= 256 floats) keeps the result live. */
int tmpnn_clock_probe(int blocks, int iters, uint64_t *out, float *sink, tmpnn_stream_t stream);

/* The clock the chip keeps under whatever runs beside this call: ONE sleeping wavefront (no LDS, it fits beside every kernel of the
 * library) reads the shader cycle counter and the 100 MHz reference around iters x s_sleep 127 (about 8 100 cycles each). Launch it on a
 * side stream while the forward runs on another: out[0] = shader cycles, out[1] = 100 MHz ticks. (The f16x2 pipeline runs power-limited:
 * 2.03-2.1 GHz against the 2.4 GHz the idle chip reports — bench.py prices its cycle figures with THIS clock.) */
int tmpnn_clock_monitor(int iters, uint64_t *out, tmpnn_stream_t stream);

/* n dependent launches of a do-nothing kernel (grid x block, lds_bytes of dynamic LDS; dirty_floats > 0: each launch writes that
 * many floats of buf): this box's price of a kernel boundary, to compare with the gaps of the real forward (tools/gap_probe.py). */
int tmpnn_launch_probe(int n, int grid, int block, int lds_bytes, float *buf, int dirty_floats, tmpnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TMPNN_DEBUG_H */
