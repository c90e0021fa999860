/*
 * vfmreg.h -- C ABI of libvfmreg_hip.so: the MI355X (gfx950) implementation of the
 * VFM-Registration correspondence-and-solve hot path.
 *
 * Conventions
 *   - plain C, no torch / C++ types; every pointer is a DEVICE pointer unless its name ends in
 *     _host; all matrices are row-major; sizes are element counts.
 *   - the caller owns every buffer.  Scratch is caller-provided and sized by the matching
 *     vfm_*_workspace_bytes(); no hidden allocation, no global state except the thread-local
 *     last-error string.  All work is enqueued on `stream` (a hipStream_t passed as void*);
 *     nothing synchronises the device unless its comment says so (vfm_voxel_robin), so the
 *     registration path can be chained and captured in a hipGraph.
 *   - nothing declared here has a process-global effect: the only state the library keeps outside the caller's buffers is THREAD-LOCAL
 *     (the last-error string, the thread's bound vfm_config_t).  The measurement hooks of the test and bench tooling (vfm_prof_*,
 *     vfm_debug_*: read-backs that synchronise the device) are NOT part of the drop-in contract: include/vfmreg_debug.h.
 *   - return 0 on success, a negative VFM_E* code otherwise; vfm_last_error() describes it.
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   RN  = src/vfm-reg/src/registration_node.py      PS  = src/vfm-reg/src/prepare_scenes.py
 *   IF  = src/vfm-reg/src/vfm_reg/image_features.py UT  = src/vfm-reg/src/vfm_reg/utils.py
 *   VHM = src/kiss-icp/cpp/kiss_icp/core/VoxelHashMap.cpp
 *   PYB = src/kiss-icp/python/kiss_icp/pybind/kiss_icp_pybind.cpp
 *   NCLT/OXF/KIT = src/vfm-reg/src/dataloader/{nclt,oxford_robotcar,kitti_odometry}.py
 */
#ifndef VFMREG_H
#define VFMREG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFM_OK 0
#define VFM_EINVAL (-1)    /* bad argument (shape, alignment, unsupported size) */
#define VFM_EWORKSPACE (-2) /* workspace too small */
#define VFM_EHIP (-3)      /* HIP runtime error at launch */

typedef void *vfm_stream_t; /* hipStream_t */

const char *vfm_last_error(void);
/* "gfx950" build id + version, for the loader's sanity check */
const char *vfm_build_info(void);

/* ------------------------------------------------------------------ kernel policy (round 6)
 * Which kernel variant / launch shape an entry point takes is a property of a CALLER-OWNED object, not of the process (SURVEY.md 8 B.5:
 * no global state except the last-error string).  A vfm_config_t starts with the factory settings -- what every call uses when nothing
 * is bound --; vfm_config_use() binds it to the CALLING THREAD (thread-local, like vfm_last_error's string; NULL unbinds) and every
 * entry point called from that thread afterwards reads its policy from it at call time.  Two pipelines with different settings are two
 * configs on two threads -- or one thread that binds the one it is about to call for.  The object must outlive its binding.
 * Every setting gives the same RESULTS (tests run the stress inputs through them); only kernels, launch shapes and times change.
 * A product integration never needs one.  Keys of vfm_config_set (value ranges as listed; unknown key: VFM_EINVAL):
 *   "coarse_slices"      map slices of the coarse pass (0 = the launchers' rules)
 *   "coarse_variant"     a code: 0 defaults; 1 / 2 = 8 waves x 32 / 4 waves x 64 resident queries; 4 = pipelined kernel with dense fp16
 *                        records; 5 = the fp16 pass in the gated family too; 7 = 5 without seed units; 12 / 10 = int8 kernel with 32 resident
 *                        queries per wave (10: two tiles per step); 20 = general selection kernel on best-score records; 21 = no chunk-major
 *                        rescan; 30 / 31 = fused fp6 half-width kernel with one (default) / two chunks per barrier; 32 / 33 = ... with two /
 *                        three (default) 32-query tiles per wave at d = 384; 40 / 41 / 42 / 43 = fp6 operand preparation by prep_chunk_kernel
 *                        (a 128-row group in registers) / prep_stream_kernel (rows read twice) / by width / prep_once_kernel (default since
 *                        round 6: one read, a tile's fp16 copy in registers); 50 / 51 = chunk-major rescan as long-lived (default) / short
 *                        workgroups; 60 / 61 = the rescan gathers its queries from the int8 fragment tiles / the row-major int8 scan (default)
 *   "match_stats"        1: the searches collect the counters vfm_debug_match_stats reads (they cost same-address atomics)
 *   "i8_min_queries"     the gated family takes the int8 pass for more than this many query rows (default 0: always)
 *   "prep_grid"          workgroups of prep_chunk_kernel: -1 (default) one per 128-row group, 0 one per compute unit, n > 0
 *   "ransac_exact_only"  1: RANSAC scores every hypothesis in fp64 (no bounds)
 *   "ransac_fused"       2 (default): the stage as 8 launches (select state reset by the centring kernel, R* out of the moment pass, the
 *                        point-wise pass from a small grid, final + mask in one workgroup); 1: 5 launches (gather + moments in one workgroup,
 *                        select + exact score per wave -- measured slower); 0: round 5's chain of 11
 *   "voxel_small"        a code: 0 / 1 = VoxelDownsample-shaped calls by the general multi-launch path / the one-launch kernel (default);
 *                        2 / 3 = round 4's / round 5's (default) per-cluster replay; 10 + k = k points per thread of the one-launch kernel
 *                        (10 = by size); 100 / 101 = its phase stamps off / on (vfm_debug_voxel_trace)
 *   "vit_gemm"           value = (narrow << 32) | (uint32) wide: narrow > 0: ViT GEMM wave tile / prefetch depth NT * 100 + PF for N <= 512
 *                        (narrow) and N > 512 (wide); narrow = -3 / -4 XCD-consistent tile mapping on (default) / off; -5 the LDS-tiled GEMM
 *                        from `wide` workgroups on (default 256, 0 never); -6 its stage shape, k-steps x 10 + stages (23); -7 attention with
 *                        K / V^T through the LDS from `wide` images on (1; 0 never); -8 waves per workgroup of the direct kernel; -9 the
 *                        token-stationary QKV / fc1 kernel from `wide` groups of 128 rows on (0 default rule, -1 never); -10 its waves per
 *                        workgroup; -11 / -12 low / high half of a device pointer to its placement trace (tools); -14 preprocessing by one
 *                        workgroup per patch (1, default) / round 1's kernel (0); -15 two (default) / one channel tile per wave; -16 timing
 *                        experiment, WRONG RESULTS; -17 residual GEMMs of the LDS-tiled path as 128 x 384 tiles (1) / 128 x 128 (0, default)
 *   "vit_fused_qkv"      QKV product + attention of an (image, head) in one workgroup (vit_qkv_attention_kernel: q, K, V^T never leave the
 *                        compute unit; the same bits as the two kernels): 0 (default) vfm_vit_forward's policy -- ViT-S width, from 24 images per
 *                        call on, unless a second round of workgroups would be less than a quarter full --, n > 0 from n images on, -1 never
 *   "vit_fused_mlp"      fc1 -> GELU -> fc2 of 128 tokens in one workgroup (vit_mlp_kernel; the same bits as the two GEMM kernels): 0 (default)
 *                        vfm_vit_forward's policy -- from two rounds of workgroups on (~150 images per call) when the last round is at least four
 *                        fifths full: a single round runs in lockstep and ends level with the two kernels --, n > 0
 *                        from n images per call on, -1 never
 *   and the single fields behind the codes: "coarse_qsets", "seed_units", "select_variant", "mx6_t4", "mx6_ns3", "prep_form" (0 .. 3),
 *   "finish_short", "rescan_rows", "vit_*", "voxel_replay2", "voxel_one_launch", "voxel_trace", "voxel_grid_ppt" (vfm_config_get reads these;
 *   cfg == NULL there: what the calling thread's entry points would read now). */
typedef struct vfm_config vfm_config_t;
int vfm_config_create(vfm_config_t **out);
int vfm_config_destroy(vfm_config_t *cfg);
int vfm_config_set(vfm_config_t *cfg, const char *key, int64_t value);
int vfm_config_get(const vfm_config_t *cfg, const char *key, int64_t *value);
int vfm_config_use(const vfm_config_t *cfg);

/* ------------------------------------------------------------------ matching (row A5) */

/* faiss::fvec_renorm_L2(d, 1, row) applied to every row (VHM:474, VHM:480): in place,
 * fp32, rows with zero norm untouched.  inv_out (nullable) receives 1/|row| (0 for zero rows).
 * Requires d % 4 == 0. */
int vfm_l2norm_rows_f32(float *x, int64_t n, int d, float *inv_out, vfm_stream_t stream);

/* precision modes of the top-1 search */
#define VFM_MATCH_FAST 0  /* fp16 MFMA coarse pass + exact fp64 re-decision (indices == EXACT) */
#define VFM_MATCH_EXACT 1 /* all-pairs fp64 on the vector ALUs (small sizes / cross-check) */

/* faiss::IndexFlatIP(d).add(m, xb) + .search(n, xq, k=1, D, I) on rows that are L2-normalised
 * first (the whole of VHM:469-495).  q, b: RAW (un-normalised) fp32 descriptors, row-major
 * n x d and m x d.  idx_out[i] = argmax_j <qn_i, bn_j> decided in fp64 on the fp32-normalised
 * rows, ties -> lowest j; sim_out[i] = (float) of that score.  Zero-norm query rows give
 * idx 0 / sim 0.  FAST requires d % 128 == 0 and d <= 768
 * (d = 640, 768 -- config C5's ViT-B/14 descriptors -- run a 4-wave kernel with 192 query registers). */
size_t vfm_match_ip_top1_workspace_bytes(int64_t n, int64_t m, int d, int prec_mode);
int vfm_match_ip_top1(const float *q, int64_t n, const float *b, int64_t m, int d, int prec_mode,
                      int64_t *idx_out, float *sim_out, void *ws, size_t ws_bytes,
                      vfm_stream_t stream);
/* one-shot form of the gated family (same workspace size; VFM_RECORDS_TOP2, the robust record kind, see below) */
int vfm_match_ip_top1_gated(const float *q, int64_t n, const float *b, int64_t m, int d, int prec_mode,
                            float gate, int64_t *idx_out, float *sim_out, void *ws, size_t ws_bytes,
                            vfm_stream_t stream);

/* Split form for a map that is searched many times (IndexFlatIP.add once, VHM:487): the
 * prepared operand holds 1/|row| and the fp16 MFMA-fragment image of the normalised rows. */
size_t vfm_match_prepared_bytes(int64_t rows, int d);
int vfm_match_prepare(const float *x, int64_t rows, int d, void *prepared, vfm_stream_t stream);
/* two operands (the map and the scan of one registration, VHM:469-482) in ONE launch */
int vfm_match_prepare2(const float *x1, int64_t rows1, void *prepared1, const float *x2, int64_t rows2,
                       void *prepared2, int d, vfm_stream_t stream);
size_t vfm_match_search_workspace_bytes(int64_t n, int64_t m, int d);
int vfm_match_search_prepared(const float *q, const void *q_prepared, int64_t n, const float *b,
                              const void *b_prepared, int64_t m, int d, int64_t *idx_out,
                              float *sim_out, void *ws, size_t ws_bytes, vfm_stream_t stream);
/* The same search as two enqueue calls, so that a caller can run the matrix-core stage of one scan
 * while the vector-ALU stage of the previous scan finishes on another stream (vfmreg/pipeline.py):
 * _coarse = fp16 MFMA pass over all N x M pairs (IndexFlatIP::search's sgemm, VHM:486-495) into ws;
 * _finish = candidate selection + exact fp64 decision from that ws.  _prepared == _coarse; _finish
 * on one stream.  ws and both prepared operands must stay untouched between the two calls. */
int vfm_match_search_coarse(const void *q_prepared, int64_t n, const void *b_prepared, int64_t m,
                            int d, void *ws, size_t ws_bytes, vfm_stream_t stream);
int vfm_match_search_finish(const float *q, const void *q_prepared, int64_t n, const float *b,
                            const void *b_prepared, int64_t m, int d, int64_t *idx_out,
                            float *sim_out, void *ws, size_t ws_bytes, vfm_stream_t stream);
/* ---- the GATED family: for a caller that keeps only matches with similarity >= gate (the cosine gate of
 * GetVFMCorrespondences, VHM:501-511).  A query whose best similarity is PROVABLY below `gate` is not resolved --
 * idx_out = -1, sim_out = -2.0 -- every other query gets the oracle's answer; gate = -INFINITY resolves every query.
 * For d = 256 ... 768 this family runs the int8 coarse pass (int8 MFMA over rows quantised per 128-row group,
 * exact integer scores, proven per-(query, chunk) bounds): twice the matrix rate of the fp16 pass, and
 * the proof that a query stays below the gate comes from the same bounds.  Elsewhere it is the ungated path and the gate
 * is ignored (every query resolved).  The three calls of one search must come from the same family:
 *   _prepare2_gated (x1 = map, x2 = scan: writes what the gated search of x2 in x1 reads -- the int8 image alone where
 *                    the int8 pass runs)  ->  _search_coarse_gated  ->  _search_finish_gated;
 * operands prepared by vfm_match_prepare / vfm_match_prepare2 carry both images and serve either family.
 * (The ungated calls above resolve every query and take the fp16 coarse pass, except for large searches -- from 8192 queries
 * x 1e9 pairs on, where the int8 pass with packed top-2 records is the faster one even without a gate; the answers are the
 * same either way.) */
int vfm_match_prepare2_gated(const float *x1, int64_t rows1, void *prepared1, const float *x2, int64_t rows2,
                             void *prepared2, int d, vfm_stream_t stream);
/* The same with the launch shape of the int8 preparation kernel chosen by the caller:
 *   VFM_PREPARE_PERSISTENT   one workgroup per compute unit, each walking several 128-row groups with the next group's rows read
 *                            under the current group's quantisation and store -- the faster form when nothing else runs
 *                            beside it or when the kernels beside it leave registers free (C2 alone: 81 vs 109 us; beside
 *                            the half-width coarse kernel: 1295 vs 1247 registrations/s);
 *   VFM_PREPARE_INTERLEAVED  one short workgroup per group -- interleaves better with a kernel whose workgroups need whole
 *                            compute units (the full-width int8 coarse kernel: 791 vs 783 registrations/s);
 *   VFM_PREPARE_DEFAULT      the library's default (INTERLEAVED). */
#define VFM_PREPARE_DEFAULT 0
#define VFM_PREPARE_PERSISTENT 1
#define VFM_PREPARE_INTERLEAVED 2
/*   VFM_PREPARE_MX6          flag, or-ed into `schedule`: write the fp6 image as well (d = 256 / 384 / 512 / 768; what the
 *                            VFM_RECORDS_MX6* searches read).  0.14 against 0.09 ms at C2 size: for d <= 384 the image is converted
 *                            from an fp16 copy of the rows inside the int8 kernel; the wider rows are read a second time by a
 *                            kernel of its own (C5 size: + 1 ms).  Ignored for other widths.  An
 *                            operand prepared WITHOUT the flag says so in its fp6 bounds (infinite): a search that asks for an
 *                            fp6 record kind on it prunes nothing and ends in the exact all-pairs decision -- the oracle's
 *                            answers, orders of magnitude slower -- rather than reading an image that is not there. */
#define VFM_PREPARE_MX6 8
/*   VFM_PREPARE_MX6_HALF     flag (implies VFM_PREPARE_MX6): the fp6 image of the FIRST d / 2 COLUMNS only -- the tile prefix the
 *                            half-width fp6 kinds (VFM_RECORDS_MX6_HALF / _HALF_FUSED) read -- and no int8 half-width image: half the
 *                            conversions, 73 MB less written at 220 000 rows x 384.  The operand's full-width fp6 bounds are then
 *                            infinite (a VFM_RECORDS_MX6 / _MX6_TOP2 search on it prunes nothing and ends in the exact decision:
 *                            correct, slow), and VFM_RECORDS_HALF / _HALF_FUSED / the half-width probe must not be run on it.
 *                            The half-width kinds bound with the image's residual over the columns they multiply either way. */
#define VFM_PREPARE_MX6_HALF 16
int vfm_match_prepare2_gated_p(const float *x1, int64_t rows1, void *prepared1, const float *x2, int64_t rows2,
                               void *prepared2, int d, int schedule, vfm_stream_t stream);
int vfm_match_search_coarse_gated(const void *q_prepared, int64_t n, const void *b_prepared, int64_t m,
                                  int d, void *ws, size_t ws_bytes, vfm_stream_t stream);
int vfm_match_search_finish_gated(const float *q, const void *q_prepared, int64_t n, const float *b,
                                  const void *b_prepared, int64_t m, int d, int64_t *idx_out,
                                  float *sim_out, void *ws, size_t ws_bytes, float gate,
                                  vfm_stream_t stream);
/* The same two calls with the kind of record the int8 pass keeps per (query, 128-row chunk) -- the same value in both:
 *   VFM_RECORDS_BEST  the best score only (the default above): the cheapest coarse kernel; every candidate chunk of a
 *                     resolved query is rescanned in int8 to find its rows (chunk-major where a map chunk collects many
 *                     queries: the int8 map is then read once; otherwise 48 KB per candidate);
 *   VFM_RECORDS_TOP2  best and second-best score plus the best row's index: ~15 % more coarse-kernel time, but a candidate
 *                     chunk with one row inside the bounds is a single 1.5 KB row -- the choice for duplicate-rich maps,
 *                     where a query has tens of candidate chunks (vfm_match_search_rescans_async reports the load);
 *   VFM_RECORDS_F16   not an int8 record kind: the fp16 coarse pass at every size, every query resolved (gate ignored) -- what
 *                     a caller falls back to on maps so duplicate-rich that the int8 bounds admit hundreds of rows per query
 *                     (operands must then carry the fp16 image: vfm_match_prepare / vfm_match_prepare2). */
#define VFM_RECORDS_BEST 0
#define VFM_RECORDS_TOP2 1
#define VFM_RECORDS_F16 2
/*   VFM_RECORDS_HALF  the half-width pass (needs a finite gate): the int8 coarse
 *                     pass over the FIRST d / 2 columns only -- half the matrix work -- with best-score records.  A (query,
 *                     chunk) pair survives if  partial score + quantisation bound + |rest of the query| * max |rest of a row
 *                     of the chunk|  can reach the gate (Cauchy-Schwarz on the other half of the columns); the survivors'
 *                     rows are scored over all d columns by the int8 rescan, and a query is resolved iff its exact best
 *                     similarity reaches the gate (every other query: idx -1, sim -2.0 -- it provably has no match).  On
 *                     descriptors whose matches stand clear of the background (the benchmark's D.2 data: 0.9 against <= 0.3)
 *                     almost nothing survives; on descriptors that are all alike everything does, and rescanning every survivor
 *                     would cost orders of magnitude more than any other mode (round 2: 171 ms at 20 000 x 200 000).  The search
 *                     guards against that ON THE DEVICE: when its bound leaves more than 48 surviving chunks per query, the same
 *                     _finish call falls through to one full-width int8 pass with the gate as hit test (match_gatepass_kernel:
 *                     6.8 ms for the whole registration at that size) -- slower than VFM_RECORDS_BEST / _TOP2 there (1.9 ms),
 *                     never a cliff, same results.  A caller that registers many scans should still probe first
 *                     (vfm_match_search_probe_half) and watch vfm_match_search_rescans_async, which reports the survivors of
 *                     every search, as vfmreg/pipeline.py does.  Exists wherever the int8 pass does (d = 256 ... 768). */
#define VFM_RECORDS_HALF 3
/*   VFM_RECORDS_HALF_FUSED  the half-width pass with its selection inside the coarse kernel: the gate is known when the coarse
 *                     pass runs (vfm_match_search_coarse_gated_g), so a (query, chunk) pair is tested the moment its best score
 *                     exists and a survivor goes straight into the chunk's rescan bin -- no records are written or read back,
 *                     no selection kernel.  The matching _finish call takes the same gate and this record kind.  Exists for
 *                     d = 256 / 384 with more than 2048 queries and at least four queries per map chunk; elsewhere it behaves
 *                     as VFM_RECORDS_HALF. */
#define VFM_RECORDS_HALF_FUSED 4
/*   VFM_RECORDS_MX6   the coarse pass over ALL d columns in microscaled fp6 (OCP MX, e2m3 elements, one power-of-two scale per
 *                     32 columns) on gfx950's scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4: twice the int8 instruction's
 *                     operations per cycle), best-score records.  The bounds are the int8 pass's with the fp6 image's MEASURED
 *                     residual norms in place of the int8 ones (about 3x wider: ~0.06 in cosine for unit Gaussian rows), so
 *                     more candidate chunks reach the int8 rescan, which -- like the fp32 refinement and the fp64 decision
 *                     behind it -- is unchanged: same answers.  Unlike VFM_RECORDS_HALF nothing here depends on how the
 *                     descriptors' energy is spread over the columns.  Needs operands prepared with VFM_PREPARE_MX6;
 *                     exists for d = 256 / 384 with more than 2048 queries, elsewhere it behaves as VFM_RECORDS_BEST. */
#define VFM_RECORDS_MX6 5
/*   VFM_RECORDS_MX6_TOP2  the fp6 pass with packed top-2 records (best and second-best score of the chunk plus the best row's
 *                     index): as VFM_RECORDS_TOP2 is to VFM_RECORDS_BEST -- a candidate chunk whose second-best row cannot reach
 *                     the bound costs one fp32 row instead of a rescan, which is what the fp6 pass's wider bounds need on
 *                     duplicate-rich maps.  Same operands and limits as VFM_RECORDS_MX6; elsewhere it behaves as VFM_RECORDS_TOP2. */
#define VFM_RECORDS_MX6_TOP2 6
/*   VFM_RECORDS_MX6_HALF  the half-width pass in fp6: VFM_RECORDS_HALF's bound -- score over the first d / 2 columns + the
 *                     image's quantisation bound + |rest of the query| * max |rest of a row of the chunk| against the gate -- on the
 *                     scaled MFMA, over the first d / 64 / 2 k-steps of the fp6 image (no second image).  Needs a finite gate
 *                     and operands prepared with VFM_PREPARE_MX6; behind the selection it is VFM_RECORDS_HALF (device-side guard
 *                     included).  d = 256 / 384 / 512 / 768 with more than 2048 queries; elsewhere it behaves as VFM_RECORDS_BEST. */
#define VFM_RECORDS_MX6_HALF 7
/*   VFM_RECORDS_MX6_HALF_FUSED  VFM_RECORDS_MX6_HALF with its selection inside the coarse kernel (needs the gate at the coarse call:
 *                     vfm_match_search_coarse_gated_g): a (query, chunk) pair whose bound reaches the gate is listed by the
 *                     kernel itself -- in the LDS, flushed once per workgroup by plain stores -- and no record is written: the
 *                     122 MB record array of a 20 000 x 200 000 search and its 70 us selection sweep do not exist.  Same
 *                     answers, same guard; where a map chunk does not collect several queries (n < 4 x chunks) it behaves as
 *                     VFM_RECORDS_MX6_HALF. */
#define VFM_RECORDS_MX6_HALF_FUSED 8
/*   VFM_RECORDS_MX6_PILOT  VFM_RECORDS_MX6 with a pilot rescan in front of the selection: the coarse kernel also notes, per query,
 *                     the chunk with the best fp6 score; the finish stage scores that ONE chunk per query exactly on the int8
 *                     image first (chunk-major, on the matrix cores) and raises the query's lower bound to what it finds --
 *                     the fp6 bound enters the selection's window once instead of twice (candidate chunks within ~0.07 of the
 *                     best cosine instead of ~0.12: about half as many on descriptors with many near neighbours, e.g. lifted
 *                     ViT features).  Costs three short launches; pays from ~8 rescanned chunks per query.  Same operands,
 *                     limits and answers as VFM_RECORDS_MX6; needs at least four queries per map chunk, else it is VFM_RECORDS_MX6. */
#define VFM_RECORDS_MX6_PILOT 9
/*   VFM_RECORDS_MX6_FUSED  VFM_RECORDS_MX6_HALF_FUSED's idea at FULL width (round 5): the fp6 pass over all d columns lists, in its
 *                     own epilogue, the (query, chunk) pairs whose best score + the image's measured bound reaches the gate, and
 *                     writes no records -- no 122 MB record array, no selection sweep.  The test is the gate alone (there is no
 *                     unmultiplied half to bound), so it prunes exactly where few rows of the map reach the gate for a query
 *                     (the planted match and nothing else); on maps where a query has many rows above the gate the
 *                     workgroups' lists overflow, the device-side guard of VFM_RECORDS_HALF goes up and one full-width int8 pass
 *                     decides every query (correct, slow) -- a caller measures (vfm_match_search_rescans_async) and uses
 *                     VFM_RECORDS_MX6 there.  Needs the gate at the coarse call (_coarse_gated_g), operands prepared with
 *                     VFM_PREPARE_MX6, d = 256 / 384, more than 2048 queries and at least four queries per map chunk; elsewhere it
 *                     behaves as VFM_RECORDS_MX6. */
#define VFM_RECORDS_MX6_FUSED 10
int vfm_match_search_coarse_gated_r(const void *q_prepared, int64_t n, const void *b_prepared, int64_t m,
                                    int d, void *ws, size_t ws_bytes, int records, vfm_stream_t stream);
/* _coarse_gated_r with the gate of the search (needed by VFM_RECORDS_HALF_FUSED; ignored by the other kinds) */
int vfm_match_search_coarse_gated_g(const void *q_prepared, int64_t n, const void *b_prepared, int64_t m,
                                    int d, void *ws, size_t ws_bytes, int records, float gate, vfm_stream_t stream);
int vfm_match_search_finish_gated_r(const float *q, const void *q_prepared, int64_t n, const float *b,
                                    const void *b_prepared, int64_t m, int d, int64_t *idx_out,
                                    float *sim_out, void *ws, size_t ws_bytes, float gate, int records,
                                    vfm_stream_t stream);
/* Round 5 -- descriptor rows stored in fp16 (BASELINE.json configs[4]: "fp16 descriptor storage"; a 1 000 000 x 768 map is 1.54 GB
 * instead of 3.07).  The reference converts its fp64 rows to fp32 one by one (VoxelHashMap.cpp:469-482) and everything behind that
 * is fp32 / fp64; here an fp16 row is widened to fp32 element by element as it is loaded -- by the preparation (norms, int8 and fp6
 * images) and by the finish stage (fp32 refinement, fp64 decision) -- and from there on the arithmetic is the fp32 path's, operation for
 * operation: the result equals the search of the widened rows, bit for bit (oracle: the same function on rows.astype(float32)).
 * _prepare2_gated_t = _prepare2_gated_p, _search_finish_gated_t = _search_finish_gated_r, each operand with its row type; the coarse
 * calls read prepared operands only and are unchanged.  fp16 rows are taken where the gated family runs its int8 / fp6 passes
 * (d = 256 ... 768); the fp16-tile pass of small searches reads fp32 rows (VFM_EINVAL otherwise). */
#define VFM_ROWS_F32 0
#define VFM_ROWS_F16 1
int vfm_match_prepare2_gated_t(const void *x1, int dtype1, int64_t rows1, void *prepared1, const void *x2, int dtype2,
                               int64_t rows2, void *prepared2, int d, int schedule, vfm_stream_t stream);
int vfm_match_search_finish_gated_t(const void *q, int dtype_q, const void *q_prepared, int64_t n, const void *b, int dtype_b,
                                    const void *b_prepared, int64_t m, int d, int64_t *idx_out, float *sim_out, void *ws,
                                    size_t ws_bytes, float gate, int records, vfm_stream_t stream);
/* Feedback for a caller that registers many scans: the number of candidate chunks the last gated search in `ws` had to
 * rescan (0 where the int8 pass did not run), copied to out_host (pinned memory) asynchronously on `stream`, after the
 * _finish_gated call on that stream.  Duplicate-rich maps put hundreds of rows inside the int8 bounds of every query;
 * beyond ~60 chunks per resolved query the ungated family (fp16 pass, 20x tighter window) is the faster one
 * (vfmreg/pipeline.py switches on this figure). */
int vfm_match_search_rescans_async(const void *ws, int64_t n, int64_t m, int32_t *out_host, vfm_stream_t stream);
/* Probe for VFM_RECORDS_HALF: runs the half-width coarse pass of this (scan, map) pair into `ws` and counts the (query, chunk)
 * pairs that survive its bound against `gate` -- nothing else is computed; the count goes to out_host (pinned memory)
 * asynchronously on `stream` (INT32_MAX, written at once, where the shape has no half-width kernel).  A half-width search
 * whose survivors run into the hundreds per query is far slower than any other mode (every survivor is a 128-row rescan), so a
 * caller probes before switching to it (vfmreg/pipeline.py: on the first registration and at every re-probe interval); `ws`
 * is free for the real search of the same pair afterwards. */
int vfm_match_search_probe_half(const void *q_prepared, int64_t n, const void *b_prepared, int64_t m, int d,
                                void *ws, size_t ws_bytes, float gate, int32_t *out_host, vfm_stream_t stream);

/* valid = !(D < min_cosine_similarity) (VHM:501-511), survivors in query order (VHM:587-600).
 * keep_out[k] = query index of the k-th survivor, *count_out = K.  corres_out (nullable,
 * n x 2 int32) receives (query index, idx[query index]) per survivor -- the Vector2iVector the
 * reference rebuilds at RN:295-317.  src_xyz_out / tgt_xyz_out (nullable, n x 3 fp64) receive
 * the coordinate pairs GetVFMCorrespondences returns (PYB:128-129); q_xyz n x 3, b_xyz m x 3. */
int vfm_threshold_compact(const float *sim, const int64_t *idx, int64_t n, double thr,
                          int64_t *keep_out, int64_t *count_out, int32_t *corres_out,
                          const double *q_xyz, const double *b_xyz, double *src_xyz_out,
                          double *tgt_xyz_out, vfm_stream_t stream);

/* find_correspondences' nearest neighbours (RN:482-538; cKDTree.query(k=1) at RN:486-496, 520-532):
 * exact Euclidean 1-NN of every row of a (n x d) among b (m x d) and, if nn_ba != NULL, of every row
 * of b among a; the decision is the fp64 squared distance accumulated in ascending k, ties -> lowest
 * index.  d2_ab (nullable): that squared distance.  Any d >= 1.
 * prec_mode VFM_MATCH_FAST (d <= 768): fp16 MFMA coarse pass on a commonly scaled copy with the norm
 * term in two appended columns (d <= 510) or in the accumulator start of each map row (wider), then the
 * fp64 decision among the candidates inside the proven error window -- same results as VFM_MATCH_EXACT
 * (all-pairs fp64), which descriptors wider than 768 fall back to.  For d = 256 ... 768 in steps of 128 the a -> b direction
 * runs the int8 coarse pass instead (map rows sorted by norm; csrc/match_l2.hip), the full b -> a direction the fp16 pass. */
size_t vfm_match_mutual_l2_workspace_bytes(int64_t n, int64_t m, int d, int prec_mode, int mutual);
int vfm_match_mutual_l2(const float *a, int64_t n, const float *b, int64_t m, int d, int prec_mode,
                        int64_t *nn_ab, double *d2_ab, int64_t *nn_ba, void *ws, size_t ws_bytes,
                        vfm_stream_t stream);

/* find_correspondences(feats0, feats1, mutual_filter=True) in ONE call (RN:482-538): the pairs (i, nn_ab[i]) whose map row's
 * nearest neighbour among a is i again (RN:520-532: `nns10[nns01] == idx0`), in ascending i.  idx0_out / idx1_out: n entries
 * each, the first *count_out valid.  nn_ab_out / d2_ab_out (nullable): the forward nearest neighbours of every row of a, as
 * vfm_match_mutual_l2 returns them.  The filter reads the reverse direction only at the matched map rows, so the reverse search
 * runs on n gathered queries instead of all m rows.  d = 256 ... 768 in steps of 128 run the int8 coarse pass both ways (map
 * rows sorted by norm, exact fp64 decision on the original rows: same pairs as the all-pairs fp64 search) -- 20 000 x 200 000 x
 * 384: see DESIGN.md 4.1 "Row A6"; other widths take vfm_match_mutual_l2's path and filter its result. */
size_t vfm_match_mutual_pairs_workspace_bytes(int64_t n, int64_t m, int d);
int vfm_match_mutual_pairs(const float *a, int64_t n, const float *b, int64_t m, int d, int64_t *idx0_out,
                           int64_t *idx1_out, int64_t *count_out, int64_t *nn_ab_out, double *d2_ab_out, void *ws,
                           size_t ws_bytes, vfm_stream_t stream);

/* ------------------------------------------------------------------ RANSAC (rows A8, A9) */

/* open3d.pipelines.registration.registration_ransac_based_on_correspondence(src, tgt, corres,
 * max_dist, TransformationEstimationPointToPoint(False), 3, RANSACConvergenceCriteria(n_iter, 1))
 * as called at RN:319-327.  src ns x 3, tgt nt x 3 fp64; corres C x 2 int32.  The number of
 * correspondences is read ON THE DEVICE from *count_dev when count_dev != NULL (<= c_max),
 * otherwise it is c_max.  Hypothesis h draws its 3 correspondences from Philox4x32-10
 * (key = seed, counter = h).  Outputs: T_out 4x4 fp64, fitness, rmse (1 fp64 each), inlier_mask
 * (c_max bytes, nullable), best_hyp (int32, -1 if no hypothesis had an inlier). */
size_t vfm_ransac_workspace_bytes(int64_t c_max, int32_t n_iter);
int vfm_ransac_corr(const double *src, const double *tgt, const int32_t *corres,
                    const int64_t *count_dev, int64_t c_max, double max_dist, int32_t n_iter,
                    uint64_t seed, double *T_out, double *fitness_out, double *rmse_out,
                    uint8_t *inlier_mask, int32_t *best_hyp_out, void *ws, size_t ws_bytes,
                    vfm_stream_t stream);
/* The same with the index check Open3D makes on the host (an out-of-range correspondence index raises there) done by the kernel that
 * gathers the point pairs: src has ns rows, tgt nt; *bad_out (device int32, ZERO before the call) becomes 1 if an index of the
 * first count rows of corres is negative or beyond its cloud -- that entry is read as row 0 and the caller discards the result (one
 * read-back with the pose instead of a min / max pass and a read-back in front of the search for the pose). */
int vfm_ransac_corr_bounded(const double *src, int64_t ns, const double *tgt, int64_t nt, const int32_t *corres,
                            const int64_t *count_dev, int64_t c_max, double max_dist, int32_t n_iter,
                            uint64_t seed, double *T_out, double *fitness_out, double *rmse_out,
                            uint8_t *inlier_mask, int32_t *best_hyp_out, int32_t *bad_out, void *ws,
                            size_t ws_bytes, vfm_stream_t stream);

/* Eigen::umeyama(with_scaling=false) / pointdsc.common.rigid_transform_3d
 * (src/vfm-reg/src/pointdsc/common.py:7-47), batched: A, B b x n x 3, w b x n or NULL,
 * denom_eps added to sum(w) in the centroids (0 for Umeyama, 1e-6 for the PointDSC variant).
 * T_out b x 16; valid (nullable) 0 where the sample is degenerate (T = identity). */
int vfm_kabsch_batched(const double *A, const double *B, const double *w, int64_t b, int64_t n,
                       double denom_eps, double *T_out, int32_t *valid, vfm_stream_t stream);

/* ------------------------------------------------------------------ projection (rows A2-A4) */

#define VFM_PROJ_NCLT 0     /* NCLT:311-366 */
#define VFM_PROJ_ROBOTCAR 1 /* OXF:330-363  */
#define VFM_PROJ_KITTI 2    /* KIT:110-125  */

/* Dataset.project_pcl_to_image.  pcl: 4 x n fp64 (row r at pcl + r*n, device); the small
 * calibration arrays are HOST pointers (copied into the launch).  mats_host: 48 fp64 =
 *   NCLT:     [T_c_body 4x4 | K 3x3 (9, rest unused) | unused]
 *   ROBOTCAR: [lidar_in_ego 4x4 | <cam>_in_ego 4x4 | inv(G_camera_image) 4x4],
 *             fc_host = fx,fy,cx,cy
 *   KITTI:    [P2 @ Tr_velo_to_cam 3x4 (12) | unused | unused]
 * win = {row0, col0, h, w} // subsample (NCLT crop window), image H x W x 3 uint8 (NCLT only,
 * nullable), outputs u, v (int32), idx (int64) in ascending point order, *count_out = K.
 * ws: vfm_project_workspace_bytes(n). */
size_t vfm_project_workspace_bytes(int64_t n);
int vfm_project_pinhole_f64(int mode, const double *pcl, int64_t n, const double *mats_host,
                            const double *fc_host, double subsample, const int64_t *win_host,
                            const uint8_t *image, int64_t H, int64_t W, int32_t *u_out,
                            int32_t *v_out, int64_t *idx_out, int64_t *count_out, void *ws,
                            size_t ws_bytes, vfm_stream_t stream);

/* F.interpolate(bilinear, align_corners=False) to Hup x Wup (IF:104-108), black-pixel zeroing
 * (PS:57-62), NCLT rot90 (PS:80-81), per-point gather feat[v,u,:] (PS:85-91) and
 * first-camera-wins scatter (PS:96-104) for ONE camera; call once per camera in priority
 * order on the same desc_out / filled (zero-initialised by the caller).
 * grid gh x gw x C fp32 (channels last); image Himg x Wimg x 3 uint8 = the UN-rotated image
 * (nullable: no black test); u, v, idx, *count_dev from vfm_project_pinhole_f64. */
int vfm_gather_bilinear_patchgrid(const float *grid, int gh, int gw, int C, int Hup, int Wup,
                                  int rot_mode, const uint8_t *image, const int32_t *u,
                                  const int32_t *v, const int64_t *idx, const int64_t *count_dev,
                                  int64_t k_max, float *desc_out, uint8_t *filled,
                                  vfm_stream_t stream);

/* create_descriptors (PS:50-107) for up to 6 cameras in ONE launch: per LiDAR point the cameras are tried
 * in array order (= the reference's dict order = priority; the first camera that sees the point wins,
 * PS:96-101), projected with the arithmetic of vfm_project_pinhole_f64, and the winning camera's patch
 * grid is sampled as in vfm_gather_bilinear_patchgrid.  Every row of desc_out (n x C) is written: zeros for a point no camera
 * sees or whose pixel is black (PS:57-62, 102-104) -- the caller need not clear it.  filled[i] = 1 iff some camera saw point i.
 * The struct array is HOST memory, its pointers DEVICE. */
typedef struct {
    int mode;                  /* VFM_PROJ_NCLT / _ROBOTCAR / _KITTI */
    double mats[48];           /* as vfm_project_pinhole_f64 */
    double fc[4];
    double subsample;
    int64_t win[4];
    int64_t H, W;              /* size of the image the projection addresses */
    const uint8_t *proj_image; /* NCLT: image of the projection's non-black test (nullable) */
    const float *grid;         /* patch grid gh x gw x C of this camera */
    const uint8_t *raw_image;  /* raw image for the black-pixel zeroing of PS:57-62 (nullable) */
    int gh, gw, Hup, Wup, rot_mode;
} vfm_lift_camera;
int vfm_lift_multicam(const double *pcl_4xn, int64_t n, int ncam, const vfm_lift_camera *cams_host,
                      int C, float *desc_out, uint8_t *filled, vfm_stream_t stream);

/* transform_pcl (UT:47-54): xyz' = T[:3,:] @ [xyz;1], fp64. T: 16 fp64 on the device. */
int vfm_transform_xyz_f64(const double *xyz, int64_t n, const double *T, double *out,
                          vfm_stream_t stream);

/* ------------------------------------------------------------------ voxel maps (row F1) */

/* kiss_icp::VoxelDownsample (src/kiss-icp/cpp/kiss_icp/core/Preprocessing.cpp:50-137; K = 1) and
 * the insertion rule of VoxelHashMap::AddPoints (VoxelHashMap.cpp:733-770, VoxelHashMap.hpp:55-62;
 * K = max_points_per_voxel): keep point i iff fewer than K earlier points lie in its voxel
 * (voxel = trunc(xyz / voxel_size)).  pts: n rows of `stride` fp64, xyz first.  keep_out: indices
 * of the survivors in ascending (input) order, *count_out their number. */
size_t vfm_voxel_first_workspace_bytes(int64_t n);
int vfm_voxel_first(const double *pts, int64_t n, int64_t stride, double voxel_size,
                    int32_t max_per_voxel, int64_t *keep_out, int64_t *count_out, void *ws,
                    size_t ws_bytes, vfm_stream_t stream);

/* The same survivors in the ORDER the reference emits them: the iteration order of the
 * tsl::robin_map<Voxel, ., VoxelHash> (Tessil robin-map v1.2.1, 3rdparty/tsl_robin/tsl_robin.cmake:24)
 * that VoxelDownsample fills and walks (Preprocessing.cpp:55-69) and that VoxelHashMap::Pointcloud /
 * PointcloudN / GetVFMCorrespondences walk (VoxelHashMap.cpp:465, 640-676); per voxel block the points
 * in insertion order.  hash_mul_y: the middle multiplier of VoxelHash -- 19349663 for
 * Preprocessing.cpp:44, 19349669 for VoxelHashMap.hpp:75.  reserve_n >= 0: the container was
 * `reserve(reserve_n)`-ed first (VoxelDownsample passes frame.size() = n); reserve_n < 0: a
 * default-constructed map that grows by doubling (VoxelHashMap::map_ / map_n_).  The chained
 * voxelisations of registration_node.py:399-414 need this order: the next level keeps the first point
 * per voxel OF THIS ORDER.  info_host (nullable, HOST int64[4]): final bucket count, number of voxels,
 * largest probe distance, generations that needed the wrap-around path.  This entry point reads the
 * voxel count back (it synchronises `stream`); it fails with VFM_EINVAL where the reference container
 * would exceed its probe-distance limit (8192) or holds a run of more than 4096 occupied buckets (its 20-bit
 * VoxelHash saturated: about 2^20 voxels and more -- the reference keeps doubling / overflows there). */
#define VFM_VOXEL_HASH_DOWNSAMPLE 19349663u
#define VFM_VOXEL_HASH_MAP 19349669u
size_t vfm_voxel_robin_workspace_bytes(int64_t n);
int vfm_voxel_robin(const double *pts, int64_t n, int64_t stride, double voxel_size,
                    int32_t max_per_voxel, uint32_t hash_mul_y, int64_t reserve_n, int64_t *keep_out,
                    int64_t *count_out, int64_t *info_host, void *ws, size_t ws_bytes,
                    vfm_stream_t stream);
/* One level of a CHAIN of VoxelDownsample()s (RN:399-414: voxel sizes .5 x, 1 x, then 5.0 m on the survivors of the one before), enqueued
 * without a read-back: the level's points are pts[idx[i]], i < *n_dev (idx NULL: pts[i]; n_dev NULL: n_max), moved by the 4 x 4 pose T_dev
 * first when given (the arithmetic of vfm_transform_xyz_f64).  keep_out: the survivors in the container's iteration order as indices INTO pts
 * -- the next level's idx; keep_local_out (nullable): their positions in this level's input; count_out: their number; info_dev (device
 * int64[8]): {buckets, voxels or -1, largest probe distance, wrapped, -, 1 = ran to its end}.  Same order as vfm_voxel_robin(..., 1,
 * hash_mul_y, n, ...) on the gathered (and moved) points, bit for bit.  The one-launch kernel only (1 <= n_max <= 2^18): a level it does
 * not reproduce (info[1] = -1: a run of occupied buckets beyond its limit; info[5] = 0: its grid did not become resident) is redone by the
 * caller through vfm_voxel_robin.  ws: vfm_voxel_robin_workspace_bytes(n_max), one per level in flight. */
int vfm_voxel_robin_level(const double *pts, int64_t stride, const int64_t *idx, int64_t n_max, const int64_t *n_dev,
                          const double *T_dev, double voxel_size, uint32_t hash_mul_y, int64_t *keep_out,
                          int64_t *keep_local_out, int64_t *count_out, int64_t *info_dev, void *ws, size_t ws_bytes,
                          vfm_stream_t stream);

/* ------------------------------------------------------------------ ICP refinement (row F2) */

/* VoxelHashMap::GetCorrespondences (src/kiss-icp/cpp/kiss_icp/core/VoxelHashMap.cpp:76-168): for
 * each source point the nearest map point among the 27 voxels around it, valid iff its distance
 * is < max_dist.  The map is a sorted-key CSR: keys[n_voxels] ascending
 * (key = ((vx+2^20)<<42)|((vy+2^20)<<21)|(vz+2^20), v = trunc(xyz / voxel_size)),
 * start[n_voxels+1], pts[start[n_voxels]][3] fp64 in (voxel, insertion) order. */
int vfm_icp_nearest(const double *src, int64_t n, const int64_t *keys, const int32_t *start,
                    const double *pts, int32_t n_voxels, double voxel_size, double max_dist,
                    double *tgt_out, uint8_t *valid_out, vfm_stream_t stream);

/* One Gauss-Newton iteration's device work in one launch: src_out = T[:3,:] @ [src; 1] (Registration.cpp:178-179 -- the
 * arithmetic of vfm_transform_xyz_f64; T_host: 16 fp64 in HOST memory, row-major, read at the call) followed by
 * vfm_icp_nearest on the moved points.  src_out may equal src. */
int vfm_icp_step_nearest(const double *src, int64_t n, const double *T_host, double *src_out, const int64_t *keys,
                         const int32_t *start, const double *pts, int32_t n_voxels, double voxel_size, double max_dist,
                         double *tgt_out, uint8_t *valid_out, vfm_stream_t stream);

/* RegisterFrame(std::vector<Eigen::VectorXd>, ...) (src/kiss-icp/cpp/kiss_icp/core/Registration.cpp:384-423): the search of its loop,
 * VoxelHashMap::GetCorrespondences(VectorXdVector) (VHM:321-448) -- among the points of the 27 voxels around a source point the first
 * minimum of |dxyz|^2 x clamp(0.5 (1 - cos(desc, desc')), 0.01, 1) (1 where a descriptor's element sum is zero), accepted if the Euclidean
 * distance is below max_dist.  vfm_icp_desc_stats: |row| and "element sum != 0" of n descriptor rows of f doubles (column order, no FMA).
 * vfm_icp_step_nearest_desc: as vfm_icp_step_nearest (T_host NULL: the points are searched where they are, as vfm_icp_nearest) with the
 * descriptor rows / statistics of the source points and of the map's points (map_*: in the order of `pts`). */
int vfm_icp_desc_stats(const double *desc, int64_t n, int32_t f, double *norm_out, uint8_t *has_out, vfm_stream_t stream);
int vfm_icp_step_nearest_desc(const double *src, int64_t n, const double *T_host, double *src_out, const double *src_desc,
                              const double *src_norm, const uint8_t *src_has, int32_t f, const int64_t *keys, const int32_t *start,
                              const double *pts, const double *map_desc, const double *map_norm, const uint8_t *map_has,
                              int32_t n_voxels, double voxel_size, double max_dist, double *tgt_out, uint8_t *valid_out,
                              vfm_stream_t stream);
/* BuildLinearSystem (src/kiss-icp/cpp/kiss_icp/core/Registration.cpp:96-141): out43 =
 * [J^T W J row-major 6x6 | J^T W r (6) | pair count], J = [I | -hat(s)], w = k^2/(k+|r|^2)^2. */
int vfm_icp_build_system(const double *src, const double *tgt, const uint8_t *valid, int64_t n,
                         double kernel, double *out43, vfm_stream_t stream);

/* ------------------------------------------------------------------ DINOv2 ViT-S/14 (row A1) */

/* self.model.model(img) of IF:101 incl. the transform of IF:67-77: bilinear resize (antialias
 * off) of B uint8 images H x W x 3 to 224 x 14*pw, ImageNet normalisation, ViT (patch 14,
 * dim = 64 * heads <= 1024 (ViT-S/14: 384, ViT-B/14: 768), 64-wide heads, LayerScale, exact GELU), final LayerNorm, cls dropped, FeatUp
 * ChannelNorm.  tokens_out: B x 16 x pw x dim fp32.  weights: packed blob described by
 * vfm_vit_weights_bytes / vfmreg/vit.py.  */
typedef struct {
    int dim;      /* 384 */
    int depth;    /* 12 */
    int heads;    /* 6 */
    int mlp_dim;  /* 1536 */
    int patch;    /* 14 */
    int patch_h;  /* 16 */
    int patch_w;  /* pw */
} vfm_vit_config;
size_t vfm_vit_weights_bytes(const vfm_vit_config *cfg);
/* segment table of the weights blob for the host-side packer: byte offsets / sizes of
 * [patch_w, patch_b, cls_pos, {qkv_w, qkv_b, qkv_c, proj_w, proj_b, ls1, fc1_w, fc1_b, fc1_c, fc2_w, fc2_b, ls2} x depth,
 *  norm_w, norm_b, channel_norm_w, channel_norm_b];
 * *_w of the linear layers are fp16 fragment tiles, everything else fp32.  The block's two LayerNorms are folded into the
 * linear layers behind them (round 4): qkv_w = W_qkv diag(ln1 gamma), qkv_b = b_qkv + W_qkv ln1_beta, qkv_c[n] = the sum of
 * row n of qkv_w AS ROUNDED to fp16; fc1_* likewise with ln2 -- the GEMM multiplies the raw residual stream and applies the
 * token's mean / 1 / std in its epilogue (csrc/vit.hip; vfmreg/vit.py does the folding).  Returns the count. */
int vfm_vit_weights_layout(const vfm_vit_config *cfg, int64_t *offsets_host, int64_t *bytes_host,
                           int max_n);
size_t vfm_vit_workspace_bytes(const vfm_vit_config *cfg, int B);
int vfm_vit_forward(const vfm_vit_config *cfg, const void *weights, const uint8_t *img, int B,
                    int H, int W, float *tokens_out, void *ws, size_t ws_bytes,
                    vfm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VFMREG_H */
