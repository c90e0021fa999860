/*
 * vfmreg_debug.h -- measurement hooks and tuning switches of libvfmreg_hip.so used by bench.py, tools/ and tests/.
 *
 * NOT part of the drop-in contract (include/vfmreg.h, SURVEY.md 8 B.5: "no global state except the last-error string").
 * Everything declared here either keeps state outside the caller's buffers or synchronises the device:
 *   - vfm_prof_*      THREAD-LOCAL, one shot: vfm_prof_arm() stores two HIP events in thread-local storage of the calling
 *                     thread; the next coarse launch issued FROM THAT THREAD (vfm_match_search_coarse* / vfm_match_ip_top1* /
 *                     vfm_match_search_prepared / vfm_match_search_probe_half) records them around its dominant kernel on the
 *                     stream it is launched on and clears the slot.  No other thread and no later search sees them.
 *   - vfm_debug_set_* PROCESS-GLOBAL A/B switches: kernel variants and launch shapes; every setting returns the same
 *                     results (tests run the stress inputs through them), only the time changes.  Not thread-safe against
 *                     concurrent searches; a product integration never calls them.
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

/* A/B: coarse-kernel variant (0 default: gated family = int8 pass for d = 256 ... 768; ungated family =
 * sparse fp16 records for d <= 384, dense fp16 records elsewhere; 1 = 8 waves x 32 queries, 2 = 4 waves x 64, 4 = pipelined
 * kernel with dense fp16 records, 5 = the fp16 pass in the gated family too, 7 = 5 without seed units, 12 = int8 kernel with
 * 32 resident queries per wave at every size, 10 = 12 with two tiles per step at every width, 20 = default kernels with the
 * general selection kernel on best-score records too, 21 = default kernels without the chunk-major rescan).
 * Values that switch one thing and leave the rest as it is (round 4): 30 / 31 = fused fp6 half-width kernel with one (default) / two
 * chunks per barrier; 32 / 33 = the same kernel at d = 384 with two / three (default since round 5) 32-query tiles per wave; 40 / 41 / 42 = fp6 operand preparation by prep_chunk_kernel (rows in registers, one pass) / prep_stream_kernel (default) / by width (d = 256 the stream form, d = 384 the one-pass form: ahead in long pipelines, behind in the 20-step form);
 * 50 / 51 = chunk-major rescan as long-lived (default) / short-lived workgroups (the fp32 refinement is one wave per list entry either way); 60 / 61 = the chunk-major rescan
 * gathers its queries from the int8 fragment tiles / from the row-major int8 scan (default) */
int vfm_debug_set_coarse_variant(int qsets);
/* A/B: force the number of map slices of the coarse pass (0 = heuristic) */
int vfm_debug_set_coarse_slices(int slices);
/* counters of the last FAST search that used workspace `ws` (candidate histogram, refined / fallback queries;
 * see csrc/match_finish.hip).  out64_host: HOST int32[64].  Synchronises the device. */
int vfm_debug_match_stats(void *ws, int64_t n, int64_t m, int32_t *out64_host);
/* the counters are collected only while this switch is on (they cost same-address atomics) */
int vfm_debug_set_match_stats(int on);
/* A/B: ViT GEMM wave tile / prefetch depth: NT * 100 + PF for N <= 512 and N > 512 (see csrc/vit.hip); narrow_cfg = -3 / -4:
 * XCD-consistent tile mapping of the ViT kernels on (default) / off; -5: the LDS-tiled GEMM from wide_cfg workgroups of 128 x 128 on
 * (0 = never, default 256); -6: its stage shape, k-steps per stage * 10 + stages (default 23); -7: attention with K / V^T shared through
 * the LDS from wide_cfg images per call on (0 = never, default 1); -8: waves per workgroup of the direct GEMM kernel (1, 2 or 4; 0 = default, 1);
 * -9: the token-stationary QKV / fc1 kernel from wide_cfg groups of 128 token rows on (0 = default: where its rounds of one workgroup per
 * compute unit are at least three quarters full; -1 = never); -10: its waves per workgroup (6, 8, 12 = default; 112 = 12 with non-temporal
 * output stores); -11 / -12: low / high 32 bits of a device pointer to its per-workgroup placement trace (tools/ab_vit_astat_trace.py; 0 = off);
 * -14: image preprocessing by one workgroup per 14 x 14 patch (1, default since round 5) / by round 1's one-thread-per-fragment-unit kernel (0);
 * -15: the token-stationary kernel with two channel tiles per wave (1, default since round 5) / one (0);
 * -16: timing experiment, WRONG RESULTS: every workgroup of the LDS-tiled kernel reads token group 0 (its A operand then hits the L2);
 * -17: the residual GEMMs (N = 384) of the LDS-tiled path as 128 x 128 tiles (0, default) / one 128 x 384 tile per workgroup (1) */
int vfm_debug_set_vit_gemm(int narrow_cfg, int wide_cfg);
/* A/B + tests: vfm_voxel_robin on a VoxelDownsample-shaped call (one point per voxel, reserve(n), n <= 2^18) by the one-launch kernel
 * (1, default: up to 256 resident workgroups with grid-wide barriers -- csrc/voxel.hip) / always by the general multi-launch path (0); both give the
 * container's order.  2 / 3: the per-cluster replay of a generation as in round 4 (a radix sort by (cluster, arrival) in front of a
 * global-memory replay) / as in round 5 (3, default: the replay sorts its cluster and runs in the LDS); 10 + k: k points per thread of
 * the one-launch kernel (10 = by size, default); 100 / 101: its phase stamps off / on (vfm_debug_voxel_trace) */
int vfm_debug_set_voxel_small(int on);
/* tools: wall-clock stamps (100 MHz) workgroup 0 of the one-launch VoxelDownsample kernel took behind each of its grid-wide barriers
 * during the last vfm_voxel_robin in `ws` (n as at that call; recorded while vfm_debug_set_voxel_small(101) is in force, 100 = off);
 * out_host: HOST int64[32], [31] = number of stamps.  Synchronises the device. */
int vfm_debug_voxel_trace(void *ws, int64_t n, int64_t *out_host);
/* A/B: workgroups of the int8 operand-preparation kernel (-1, default = one per 128-row group; 0 = one per compute unit, each
 * walking several groups with the next group's rows read under the current group's quantisation and store: faster alone,
 * slower beside the coarse kernel; n > 0 = n workgroups) */
int vfm_debug_set_prep_grid(int workgroups);
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
/* A/B: the gated family takes the int8 pass for more than this many query rows (default 0: always) */
int vfm_debug_set_i8_min_queries(int n);
/* tests / bench: what the last vfm_ransac_corr in `ws` (same c_max, n_iter) did: out_host[0] = hypotheses scored in fp64 from the candidate
 * list, [1] = the list overflowed (every hypothesis scored in fp64), [2] = the point-wise fp32 pass was needed.  Synchronises. */
int vfm_debug_ransac_counts(const void *ws, int64_t c_max, int32_t n_iter, int32_t *out_host);
/* A/B: 1 = RANSAC scores every hypothesis in fp64 (skips the bounds) */
int vfm_debug_set_ransac_exact_only(int on);

#ifdef __cplusplus
}
#endif
#endif /* VFMREG_DEBUG_H */
