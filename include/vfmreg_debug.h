/*
 * vfmreg_debug.h -- measurement hooks and tuning switches of libvfmreg_hip.so used by bench.py, tools/ and tests/.
 *
 * NOT part of the drop-in contract (include/vfmreg.h, SURVEY.md 8 B.5: "no global state except the last-error string").
 * Everything declared here either keeps state outside the caller's buffers or synchronises the device:
 *   - vfm_prof_*      THREAD-LOCAL, one shot: vfm_prof_arm() stores two HIP events in thread-local storage of the calling
 *                     thread; the next coarse launch issued FROM THAT THREAD (vfm_match_search_coarse* / vfm_match_ip_top1* /
 *                     vfm_match_search_prepared / vfm_match_search_probe_half) records them around its dominant kernel on the
 *                     stream it is launched on and clears the slot.  No other thread and no later search sees them.
 *   - (until round 5 this header also held vfm_debug_set_*: PROCESS-GLOBAL A/B switches.  They are gone: kernel policy is a caller-owned
 *     vfm_config_t bound per thread -- include/vfmreg.h, vfm_config_* -- and vfmreg/_lib.py keeps the old names as Python functions that
 *     set the calling thread's config, for the tools.)
 *   - vfm_debug_match_stats / vfm_debug_i8_rows / vfm_debug_mx6_rows  read-backs for tests; they synchronise the device.
 */
#ifndef VFMREG_DEBUG_H
#define VFMREG_DEBUG_H

#include "vfmreg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HIP events around the dominant kernel (the MFMA coarse pass of the top-1 search), recorded on the stream that kernel is
 * launched on.  vfm_prof_arm() applies to the NEXT search issued from the calling thread (one shot, thread-local).
 * vfm_prof_elapsed_ms() waits for `stop`. */
int vfm_prof_events_create(void **start, void **stop);
int vfm_prof_arm(void *start, void *stop);
int vfm_prof_elapsed_ms(void *start, void *stop, float *ms_host);
int vfm_prof_events_destroy(void *start, void *stop);

/* counters of the last FAST search that used workspace `ws` (candidate histogram, refined / fallback queries;
 * see csrc/match_finish.hip).  out64_host: HOST int32[64].  Synchronises the device. */
int vfm_debug_match_stats(void *ws, int64_t n, int64_t m, int32_t *out64_host);
/* tools: wall-clock stamps (100 MHz) workgroup 0 of the one-launch VoxelDownsample kernel took behind each of its grid-wide barriers
 * during the last vfm_voxel_robin in `ws` (n as at that call; recorded while vfm_debug_set_voxel_small(101) is in force, 100 = off);
 * out_host: HOST int64[32], [31] = number of stamps.  Synchronises the device. */
int vfm_debug_voxel_trace(void *ws, int64_t n, int64_t *out_host);
/* tests: the int8 image of a prepared operand (d = 256, 384) unpacked on the host -- q8_host[rows][d], and per row the
 * quantisation step of its 128-row group, its residual norm E and the group's maximum E.  Synchronises the device. */
int vfm_debug_i8_rows(const void *prepared, int64_t rows, int d, int8_t *q8_host, float *step_host,
                      float *err_host, float *gerr_host);
/* tests: the fp6 image of an operand prepared with VFM_PREPARE_MX6 (d = 256, 384), dequantised on the host -- v6_host[rows][d]
 * (code value x block scale), and per row the image's residual norm E (arithmetic slack included) and its group's maximum.
 * Synchronises the device. */
int vfm_debug_mx6_rows(const void *prepared, int64_t rows, int d, float *v6_host, float *err_host, float *gerr_host);
/* tests: the same image's residual norm over the first d / 2 columns (the bound of the half-width fp6 kinds) and its group maximum */
int vfm_debug_mx6_half_err(const void *prepared, int64_t rows, int d, float *errh_host, float *gerrh_host);
/* tests / bench: what the last vfm_ransac_corr in `ws` (same c_max, n_iter) did: out_host[0] = hypotheses scored in fp64 from the candidate
 * list, [1] = the list overflowed (every hypothesis scored in fp64), [2] = the point-wise fp32 pass was needed.  Synchronises. */
int vfm_debug_ransac_counts(const void *ws, int64_t c_max, int32_t n_iter, int32_t *out_host);

#ifdef __cplusplus
}
#endif
#endif /* VFMREG_DEBUG_H */
