// match_api.hip -- the matcher's C ABI (include/vfmreg.h): which coarse pass and record kind a search takes, the argument
// block of the coarse kernels, workspace layout queries, the one-shot and split entry points, experiment knobs and the
// profiling hook.  Kernels: match_prep.hip, match_coarse_f16.hip, match_coarse_i8.hip, match_finish.hip, match_l2.hip.
#include "match_internal.h"

namespace vfmm {


int choose_slices(int nqb, int nchunks) {
    if (vfm_cfg().force_slices > 0) return vfm_cfg().force_slices < nchunks ? vfm_cfg().force_slices : nchunks;
    // Fill 256 CUs with whole "rounds" of workgroups (tail efficiency).  Every (query block, slice) unit re-reads its 256
    // queries (196 KB), so HBM / Infinity-Cache traffic grows with the slice count (C2: 55 slices 1.45 GB per launch).
    // Round 2 sweep at C2 with the seeded sparse kernel (registrations/s in the pipeline): 14-16 slices 349, 21: 367,
    // 29: 366, 35: 367, 42: 365, 48: 364, 55: 364 -- flat from ~8 rounds on.  So: among the counts within 1 % of the best
    // tail efficiency take the SMALLEST that still gives >= 8 rounds (29 at C2); without such a count, the most efficient.
    int best_s = 1;
    double best_eff = -1.0;
    const int smax = nchunks < 64 ? nchunks : 64;
    auto eff_of = [&](int s, long long* rounds_out) {
        const long long total = (long long)nqb * s;
        const long long rounds = (total + 255) / 256;
        if (rounds_out) *rounds_out = rounds;
        return (double)total / (double)(rounds * 256);
    };
    for (int s = 1; s <= smax; ++s) {
        if (nchunks / s < 8 && s > 1) break;  // keep >= 32 tiles per workgroup
        const double eff = eff_of(s, nullptr);
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best_s = s;
        }
    }
    for (int s = 1; s < best_s; ++s) {
        long long rounds;
        const double eff = eff_of(s, &rounds);
        if (rounds >= 8 && eff >= best_eff - 0.01) return s;
    }
    return best_s;
}

// profiling hook (vfm_prof_arm): events recorded around the next coarse launch on this thread
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

// 0 = default (pipelined kernel for d <= 384); 1 = match_coarse_kernel<.,1>; 2 = match_coarse_kernel<.,2>;
// set through vfm_debug_set_coarse_variant for A/B runs
// 4 = pipelined kernel with the DENSE per-chunk records + match_select_kernel (round-1 path; A/B reference)

// inner-product searches with d <= 384 use the sparse row-level records of match_coarse_pipe_kernel<., true>
// (from 3 query blocks on: with 1-2 blocks every unit of the grid runs at once, none is seeded, and the dense records
// measured faster -- 244 vs 282 us per registration at 300 x 50000)
bool use_sparse(int d, int64_t n, int64_t m) {
    return d <= 384 && d % 128 == 0 && m < (1ll << 24) && n > 2 * QBLOCK &&
           (vfm_cfg().coarse_qsets == 0 || vfm_cfg().coarse_qsets == 3 || vfm_cfg().coarse_qsets == 5);
}

// ... and before those, in the GATED family of entry points (callers that keep only matches above a similarity gate), for
// d = 256 / 384: the int8 coarse pass with one best-score record per (query, chunk).  Its exact re-decision costs an int8
// rescan per candidate chunk, cheap when most unmatched queries stop at the gate and slower than the fp16 pass when every
// query must be resolved -- so the ungated entry points keep the fp16 pass.  (variant 5 = fp16 pass everywhere, A/B)
// Ungated calls (every query resolved) take the int8 pass, with packed top-2 records, from 8192 queries x 1e9 pairs on
// (tools/time_ungated.py, coarse + finish: 20 000 x 200 000 x 384 1.98 vs 2.69 ms, 20 000 x 50 000 x 256 0.55 vs 0.65,
// 50 000 x 1 000 000 x 768 41.8 vs 74.2; below that the fp16 pass wins: 2000 x 200 000 0.39 vs 0.45, 300 x 50 000 0.15 vs 0.22).
bool use_i8(int d, int64_t n, int64_t m, bool gated) {
    const bool large = n >= 8192 && n * m >= 1000000000ll;
    return (gated || large) && i8_capable(d) && m < (1ll << 24) && n > vfm_cfg().i8_min_queries &&
           (vfm_cfg().coarse_qsets == 0 || vfm_cfg().coarse_qsets == 10 || vfm_cfg().coarse_qsets == 12);
}

// queries per workgroup of the coarse kernel that do_search_coarse will launch
int coarse_qblock(int d) { return d > 512 ? 128 : QBLOCK; }

CoarseArgs coarse_args(const Prepared& Q, const Prepared& B, const SearchWs& w, int64_t n, int64_t m, int qblock) {
    const int64_t npad = rows_padded(n), mpad = rows_padded(m);
    CoarseArgs a;
    a.Qh = Q.tiles;
    a.Bh = B.tiles;
    a.partials = w.partials;
    a.nq_tiles = (int)((n + TILE_ROWS - 1) / TILE_ROWS);
    a.nchunks = (int)(mpad / CHUNK_ROWS);
    a.m_valid = m;
    a.npad = (int)npad;
    a.nqb = (int)(npad / qblock);
    a.nslices = choose_slices(a.nqb, a.nchunks);
    a.qmax = w.qmax;
    a.first_pad_chunk = (int)(m / CHUNK_ROWS);
    a.row_bias = nullptr;
    a.qinv = Q.inv;
    a.seed_parts = a.seed_chunks = a.nseed_pad = 0;
    a.rec_cnt = nullptr;
    a.rec = nullptr;
    a.rcap = 0;
    a.window = DEFAULT_WINDOW;
    a.ib = I8Bounds{nullptr, nullptr, nullptr, nullptr, 0};
    a.qbest = nullptr;
    return a;
}

// stage 1 of a search: the MFMA coarse pass (fills ws: partials + per-query coarse maxima)
// records (int8 pass): 0 = best score per (query, chunk), 1 = packed top-2 with the best row's index (VFM_RECORDS_*)
int do_search_coarse(const void* qprep, int64_t n, const void* bprep, int64_t m, int d, void* ws, hipStream_t st,
                     bool bias_from_map_inv, bool inner_product, bool gated, int records, float gate) {
    Prepared Q = carve_prepared(const_cast<void*>(qprep), n, d);
    Prepared B = carve_prepared(const_cast<void*>(bprep), m, d);
    SearchWs w = carve_search(ws, n, m);
    CoarseArgs a = coarse_args(Q, B, w, n, m, coarse_qblock(d));
    if (bias_from_map_inv) {  // Euclidean search, d > 510: the map's "inv" array holds -|b~|^2 / 2
        if (d != 640 && d != 768) return vfm_fail(VFM_EINVAL, "row bias needs the 4-wave coarse kernel (K = 640 / 768), got %d", d);
        a.row_bias = B.inv;
    }
    const bool i8 = inner_product && records != VFM_RECORDS_F16 && use_i8(d, n, m, gated);
    if (inner_product && !i8 && use_sparse(d, n, m)) {
        a.rec_cnt = w.rec_cnt;
        a.rec = w.rec;
        a.rcap = w.rcap;
        if (vfm_cfg().seed_units && a.nchunks >= 256 && a.nqb <= 256) {  // seed units: one short round at the head of the grid
            a.seed_parts = 256 / a.nqb < 4 ? 256 / a.nqb : 4;
            a.seed_chunks = 5;
            a.nseed_pad = (a.nqb * a.seed_parts + 7) / 8 * 8;
            a.nslices = choose_slices(a.nqb, a.nchunks - a.seed_parts * a.seed_chunks);
        }
    }
    VFM_CHECK_HIP(hipMemsetAsync(w.fb_count, 0, search_zero_bytes(n, m), st));  // fb_count | qmax | rec_cnt | bin_cnt
    if (i8) {
        if (!gated) records = VFM_RECORDS_TOP2;  // no feedback loop behind an ungated call: the robust record kind
        records = effective_records(records, d, n, m);
        if (records == VFM_RECORDS_HALF_FUSED || records == VFM_RECORDS_MX6_HALF_FUSED || records == VFM_RECORDS_MX6_FUSED) {
            if (!(gate > -__builtin_inff())) return vfm_fail(VFM_EINVAL, "search_coarse: VFM_RECORDS_HALF_FUSED needs a finite gate");
            a.gate = gate;
            a.n_valid = n;
            a.qrest = Q.rest;
            a.grest = B.grest;
            a.bin_cnt = w.bin_cnt;
            a.bins = w.bins;
            a.bin_cap = w.bin_cap;
            a.cand_cnt = w.cand_cnt;
            a.cand = w.cand;
            a.cap = w.cap;
            a.survivors = w.fb_count + 5;
            a.surv = reinterpret_cast<unsigned*>(w.partials);   // the fp6 form: one slot per workgroup in the record buffer it does not write
            VFM_CHECK_HIP(hipMemsetAsync(w.cand_cnt, 0, (size_t)a.npad * sizeof(int), st));  // lengths of the queries' own lists
        }
        const bool half = records == VFM_RECORDS_HALF || records == VFM_RECORDS_HALF_FUSED;   // the image of the first d / 2 columns
        a.Qh = half ? Q.tiles8h : Q.tiles8;
        a.Bh = half ? B.tiles8h : B.tiles8;
        a.ib = I8Bounds{Q.err, Q.gstep, B.gstep, B.gerr, records == VFM_RECORDS_TOP2 ? 1 : 0};
        if (records == VFM_RECORDS_MX6_PILOT) {   // the full-width fp6 pass, which also notes every query's best chunk
            a.qbest = w.qbest;
            records = VFM_RECORDS_MX6;
        }
        if (records == VFM_RECORDS_MX6 || records == VFM_RECORDS_MX6_TOP2 || records == VFM_RECORDS_MX6_HALF || records == VFM_RECORDS_MX6_HALF_FUSED ||
            records == VFM_RECORDS_MX6_FUSED) {
            // the fp6 image and its bounds (operands prepared with VFM_PREPARE_MX6)
            a.Qh = Q.tiles6;
            a.Bh = B.tiles6;
            const bool fuse6 = records == VFM_RECORDS_MX6_HALF_FUSED, fusefull = records == VFM_RECORDS_MX6_FUSED;
            a.ib = (fuse6 || records == VFM_RECORDS_MX6_HALF) ? mx6_bounds_half(Q, B) : mx6_bounds(Q, B, records == VFM_RECORDS_MX6_TOP2 ? 1 : 0);
            return launch_coarse_mx6(a, d, records == VFM_RECORDS_MX6_TOP2, records == VFM_RECORDS_MX6_HALF || fuse6, fuse6 || fusefull, st);
        }
        return launch_coarse_int8(a, d, n, records, st);
    }
    return launch_coarse_f16(a, d, st);
}

int do_search(const float* q, const void* qprep, int64_t n, const float* b, const void* bprep, int64_t m, int d,
              int64_t* idx_out, float* sim_out, void* ws, hipStream_t st) {
    const int rc = do_search_coarse(qprep, n, bprep, m, d, ws, st, false, true);
    if (rc) return rc;
    return do_search_finish(q, qprep, n, b, bprep, m, d, idx_out, sim_out, ws, st);
}


}  // namespace vfmm

using namespace vfmm;

VFM_EXPORT size_t vfm_match_prepared_bytes(int64_t rows, int d) { return carve_prepared(nullptr, rows, d).bytes; }

VFM_EXPORT int vfm_match_prepare(const float* x, int64_t rows, int d, void* prepared, vfm_stream_t stream) {
    VFM_CHECK_ARG(rows > 0 && d % 128 == 0 && d >= 128 && d <= 768, "prepare: d must be in {128,256,384,512,640,768}");
    VFM_CHECK_ARG(x && prepared, "prepare: null pointer");
    return do_prepare(x, rows, d, prepared, (hipStream_t)stream);
}

VFM_EXPORT int vfm_match_prepare2(const float* x1, int64_t rows1, void* prepared1, const float* x2, int64_t rows2, void* prepared2,
                                  int d, vfm_stream_t stream) {
    VFM_CHECK_ARG(rows1 > 0 && rows2 > 0 && d % 128 == 0 && d >= 128 && d <= 768, "prepare2: d must be in {128,256,384,512,640,768}");
    VFM_CHECK_ARG(x1 && x2 && prepared1 && prepared2, "prepare2: null pointer");
    return do_prepare2(x1, rows1, prepared1, x2, rows2, prepared2, d, (hipStream_t)stream);
}

VFM_EXPORT int vfm_match_prepare2_gated(const float* x1, int64_t rows1, void* prepared1, const float* x2, int64_t rows2,
                                        void* prepared2, int d, vfm_stream_t stream) {
    VFM_CHECK_ARG(rows1 > 0 && rows2 > 0 && d % 128 == 0 && d >= 128 && d <= 768, "prepare2: d must be in {128,256,384,512,640,768}");
    VFM_CHECK_ARG(x1 && x2 && prepared1 && prepared2, "prepare2: null pointer");
    // (map, scan): where the gated search of x2 in x1 runs the int8 pass, the fp16 image is never read
    const bool want_f16 = !use_i8(d, rows2, rows1, true);
    return do_prepare2(x1, rows1, prepared1, x2, rows2, prepared2, d, (hipStream_t)stream, want_f16);
}

VFM_EXPORT int vfm_match_prepare2_gated_p(const float* x1, int64_t rows1, void* prepared1, const float* x2, int64_t rows2,
                                          void* prepared2, int d, int schedule, vfm_stream_t stream) {
    VFM_CHECK_ARG(rows1 > 0 && rows2 > 0 && d % 128 == 0 && d >= 128 && d <= 768, "prepare2: d must be in {128,256,384,512,640,768}");
    VFM_CHECK_ARG(x1 && x2 && prepared1 && prepared2, "prepare2: null pointer");
    VFM_CHECK_ARG((schedule & ~(VFM_PREPARE_MX6 | VFM_PREPARE_MX6_HALF)) >= VFM_PREPARE_DEFAULT &&
                      (schedule & ~(VFM_PREPARE_MX6 | VFM_PREPARE_MX6_HALF)) <= VFM_PREPARE_INTERLEAVED,
                  "prepare2: unknown schedule %d", schedule);
    const bool want_f16 = !use_i8(d, rows2, rows1, true);
    return do_prepare2(x1, rows1, prepared1, x2, rows2, prepared2, d, (hipStream_t)stream, want_f16, schedule);
}

// the gated pair with the rows' storage type stated (VFM_ROWS_F32 / VFM_ROWS_F16): fp16 rows are widened to fp32 element by element as
// they are loaded, everything else is the fp32 path
VFM_EXPORT int vfm_match_prepare2_gated_t(const void* x1, int dtype1, int64_t rows1, void* prepared1, const void* x2, int dtype2,
                                          int64_t rows2, void* prepared2, int d, int schedule, vfm_stream_t stream) {
    VFM_CHECK_ARG(rows1 > 0 && rows2 > 0 && d % 128 == 0 && d >= 128 && d <= 768, "prepare2: d must be in {128,256,384,512,640,768}");
    VFM_CHECK_ARG(x1 && x2 && prepared1 && prepared2, "prepare2: null pointer");
    VFM_CHECK_ARG((dtype1 == VFM_ROWS_F32 || dtype1 == VFM_ROWS_F16) && (dtype2 == VFM_ROWS_F32 || dtype2 == VFM_ROWS_F16), "prepare2: unknown row type");
    VFM_CHECK_ARG((schedule & ~(VFM_PREPARE_MX6 | VFM_PREPARE_MX6_HALF)) >= VFM_PREPARE_DEFAULT &&
                      (schedule & ~(VFM_PREPARE_MX6 | VFM_PREPARE_MX6_HALF)) <= VFM_PREPARE_INTERLEAVED,
                  "prepare2: unknown schedule %d", schedule);
    const bool want_f16 = !use_i8(d, rows2, rows1, true);
    return do_prepare2(Rows(x1, dtype1 == VFM_ROWS_F16), rows1, prepared1, Rows(x2, dtype2 == VFM_ROWS_F16), rows2, prepared2, d,
                       (hipStream_t)stream, want_f16, schedule);
}

VFM_EXPORT size_t vfm_match_search_workspace_bytes(int64_t n, int64_t m, int d) {
    (void)d;
    return carve_search(nullptr, n, m).bytes;
}

VFM_EXPORT int vfm_match_search_prepared(const float* q, const void* q_prepared, int64_t n, const float* b,
                                         const void* b_prepared, int64_t m, int d, int64_t* idx_out, float* sim_out,
                                         void* ws, size_t ws_bytes, vfm_stream_t stream) {
    VFM_CHECK_ARG(n > 0 && m > 0, "search: empty operand (n=%lld m=%lld)", (long long)n, (long long)m);
    VFM_CHECK_ARG(d % 128 == 0 && d >= 128 && d <= 768, "search: d must be in {128,256,384,512,640,768}");
    VFM_CHECK_ARG(m < (1ll << 31) - 256 && n < (1ll << 31) - 256, "search: more than 2^31 rows");
    if (ws_bytes < vfm_match_search_workspace_bytes(n, m, d)) return vfm_fail(VFM_EWORKSPACE, "search: workspace too small");
    return do_search(q, q_prepared, n, b, b_prepared, m, d, idx_out, sim_out, ws, (hipStream_t)stream);
}

static int check_search_args(int64_t n, int64_t m, int d, size_t ws_bytes) {
    VFM_CHECK_ARG(n > 0 && m > 0, "search: empty operand (n=%lld m=%lld)", (long long)n, (long long)m);
    VFM_CHECK_ARG(d % 128 == 0 && d >= 128 && d <= 768, "search: d must be in {128,256,384,512,640,768}");
    VFM_CHECK_ARG(m < (1ll << 31) - 256 && n < (1ll << 31) - 256, "search: more than 2^31 rows");
    if (ws_bytes < vfm_match_search_workspace_bytes(n, m, d)) return vfm_fail(VFM_EWORKSPACE, "search: workspace too small");
    return VFM_OK;
}

VFM_EXPORT int vfm_match_search_coarse(const void* q_prepared, int64_t n, const void* b_prepared, int64_t m, int d,
                                       void* ws, size_t ws_bytes, vfm_stream_t stream) {
    if (int rc = check_search_args(n, m, d, ws_bytes)) return rc;
    VFM_CHECK_ARG(q_prepared && b_prepared && ws, "search_coarse: null pointer");
    return do_search_coarse(q_prepared, n, b_prepared, m, d, ws, (hipStream_t)stream, false, true);
}

VFM_EXPORT int vfm_match_search_finish(const float* q, const void* q_prepared, int64_t n, const float* b,
                                       const void* b_prepared, int64_t m, int d, int64_t* idx_out, float* sim_out,
                                       void* ws, size_t ws_bytes, vfm_stream_t stream) {
    if (int rc = check_search_args(n, m, d, ws_bytes)) return rc;
    VFM_CHECK_ARG(q && b && q_prepared && b_prepared && ws && idx_out && sim_out, "search_finish: null pointer");
    return do_search_finish(q, q_prepared, n, b, b_prepared, m, d, idx_out, sim_out, ws, (hipStream_t)stream);
}

VFM_EXPORT int vfm_match_search_coarse_gated(const void* q_prepared, int64_t n, const void* b_prepared, int64_t m, int d,
                                             void* ws, size_t ws_bytes, vfm_stream_t stream) {
    return vfm_match_search_coarse_gated_r(q_prepared, n, b_prepared, m, d, ws, ws_bytes, VFM_RECORDS_BEST, stream);
}

VFM_EXPORT int vfm_match_search_coarse_gated_r(const void* q_prepared, int64_t n, const void* b_prepared, int64_t m, int d,
                                               void* ws, size_t ws_bytes, int records, vfm_stream_t stream) {
    if (int rc = check_search_args(n, m, d, ws_bytes)) return rc;
    VFM_CHECK_ARG(q_prepared && b_prepared && ws, "search_coarse: null pointer");
    VFM_CHECK_ARG((records >= VFM_RECORDS_BEST && records <= VFM_RECORDS_HALF) || (records >= VFM_RECORDS_MX6 && records <= VFM_RECORDS_MX6_HALF) || records == VFM_RECORDS_MX6_PILOT, "search_coarse: unknown record kind %d", records);
    return do_search_coarse(q_prepared, n, b_prepared, m, d, ws, (hipStream_t)stream, false, true, true, records);
}

VFM_EXPORT int vfm_match_search_coarse_gated_g(const void* q_prepared, int64_t n, const void* b_prepared, int64_t m, int d,
                                               void* ws, size_t ws_bytes, int records, float gate, vfm_stream_t stream) {
    if (int rc = check_search_args(n, m, d, ws_bytes)) return rc;
    VFM_CHECK_ARG(q_prepared && b_prepared && ws, "search_coarse: null pointer");
    VFM_CHECK_ARG(records >= VFM_RECORDS_BEST && records <= VFM_RECORDS_MX6_FUSED, "search_coarse: unknown record kind %d", records);
    VFM_CHECK_ARG(gate == gate, "search_coarse: gate is NaN");
    return do_search_coarse(q_prepared, n, b_prepared, m, d, ws, (hipStream_t)stream, false, true, true, records, gate);
}

VFM_EXPORT int vfm_match_search_finish_gated(const float* q, const void* q_prepared, int64_t n, const float* b,
                                             const void* b_prepared, int64_t m, int d, int64_t* idx_out, float* sim_out,
                                             void* ws, size_t ws_bytes, float gate, vfm_stream_t stream) {
    return vfm_match_search_finish_gated_r(q, q_prepared, n, b, b_prepared, m, d, idx_out, sim_out, ws, ws_bytes, gate,
                                           VFM_RECORDS_BEST, stream);
}

VFM_EXPORT int vfm_match_search_finish_gated_r(const float* q, const void* q_prepared, int64_t n, const float* b,
                                               const void* b_prepared, int64_t m, int d, int64_t* idx_out, float* sim_out,
                                               void* ws, size_t ws_bytes, float gate, int records, vfm_stream_t stream) {
    if (int rc = check_search_args(n, m, d, ws_bytes)) return rc;
    VFM_CHECK_ARG(q && b && q_prepared && b_prepared && ws && idx_out && sim_out, "search_finish: null pointer");
    VFM_CHECK_ARG(gate == gate, "search_finish: gate is NaN");
    VFM_CHECK_ARG(records >= VFM_RECORDS_BEST && records <= VFM_RECORDS_MX6_FUSED, "search_finish: unknown record kind %d", records);
    return do_search_finish(q, q_prepared, n, b, b_prepared, m, d, idx_out, sim_out, ws, (hipStream_t)stream, true, gate, records);
}

VFM_EXPORT int vfm_match_search_finish_gated_t(const void* q, int dtype_q, const void* q_prepared, int64_t n, const void* b, int dtype_b,
                                               const void* b_prepared, int64_t m, int d, int64_t* idx_out, float* sim_out, void* ws,
                                               size_t ws_bytes, float gate, int records, vfm_stream_t stream) {
    if (int rc = check_search_args(n, m, d, ws_bytes)) return rc;
    VFM_CHECK_ARG(q && b && q_prepared && b_prepared && ws && idx_out && sim_out, "search_finish: null pointer");
    VFM_CHECK_ARG(gate == gate, "search_finish: gate is NaN");
    VFM_CHECK_ARG(records >= VFM_RECORDS_BEST && records <= VFM_RECORDS_MX6_FUSED, "search_finish: unknown record kind %d", records);
    VFM_CHECK_ARG((dtype_q == VFM_ROWS_F32 || dtype_q == VFM_ROWS_F16) && (dtype_b == VFM_ROWS_F32 || dtype_b == VFM_ROWS_F16), "search_finish: unknown row type");
    VFM_CHECK_ARG((dtype_q == VFM_ROWS_F32 && dtype_b == VFM_ROWS_F32) || (records != VFM_RECORDS_F16 && use_i8(d, n, m, true)),
                  "search_finish: fp16 rows are taken by the int8 / fp6 searches only");
    return do_search_finish(Rows(q, dtype_q == VFM_ROWS_F16), q_prepared, n, Rows(b, dtype_b == VFM_ROWS_F16), b_prepared, m, d, idx_out, sim_out,
                            ws, (hipStream_t)stream, true, gate, records);
}

VFM_EXPORT int vfm_match_search_rescans_async(const void* ws, int64_t n, int64_t m, int32_t* out_host, vfm_stream_t stream) {
    VFM_CHECK_ARG(ws && out_host && n > 0 && m > 0, "search_rescans: bad arguments");
    SearchWs w = carve_search(const_cast<void*>(ws), n, m);
    VFM_CHECK_HIP(hipMemcpyAsync(out_host, w.fb_count + 5, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return VFM_OK;
}

VFM_EXPORT int vfm_match_search_probe_half(const void* q_prepared, int64_t n, const void* b_prepared, int64_t m, int d, void* ws,
                                           size_t ws_bytes, float gate, int32_t* out_host, vfm_stream_t stream) {
    if (int rc = check_search_args(n, m, d, ws_bytes)) return rc;
    VFM_CHECK_ARG(q_prepared && b_prepared && ws && out_host, "probe_half: null pointer");
    VFM_CHECK_ARG(gate > -__builtin_inff(), "probe_half: needs a finite gate");
    if (!(use_i8(d, n, m, true) && half_capable(d, n))) {  // no half-width kernel for this shape: "everything survives"
        *out_host = INT32_MAX;
        return VFM_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    if (int rc = do_search_coarse(q_prepared, n, b_prepared, m, d, ws, st, false, true, true, VFM_RECORDS_HALF)) return rc;
    if (int rc = probe_half_select(q_prepared, n, b_prepared, m, d, ws, gate, st)) return rc;
    SearchWs w = carve_search(ws, n, m);
    VFM_CHECK_HIP(hipMemcpyAsync(out_host, w.fb_count + 5, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    return VFM_OK;
}

VFM_EXPORT size_t vfm_match_ip_top1_workspace_bytes(int64_t n, int64_t m, int d, int prec_mode) {
    if (prec_mode == VFM_MATCH_EXACT) return vfm_align_up((size_t)(n + m) * sizeof(float), 256) + 512;
    return vfm_match_prepared_bytes(n, d) + vfm_match_prepared_bytes(m, d) + vfm_match_search_workspace_bytes(n, m, d);
}

static int ip_top1(const float* q, int64_t n, const float* b, int64_t m, int d, int prec_mode, bool gated, float gate, int64_t* idx_out,
            float* sim_out, void* ws, size_t ws_bytes, vfm_stream_t stream) {
    VFM_CHECK_ARG(n > 0 && m > 0, "match: empty operand (n=%lld m=%lld)", (long long)n, (long long)m);
    VFM_CHECK_ARG(q && b && idx_out && sim_out && ws, "match: null pointer");
    if (ws_bytes < vfm_match_ip_top1_workspace_bytes(n, m, d, prec_mode)) return vfm_fail(VFM_EWORKSPACE, "match: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (prec_mode == VFM_MATCH_EXACT) {
        VFM_CHECK_ARG(d % 4 == 0 && d <= 1024, "match(EXACT): need d %% 4 == 0 and d <= 1024");
        return exact_ip_top1(q, n, b, m, d, idx_out, sim_out, ws, st);
    }
    VFM_CHECK_ARG(prec_mode == VFM_MATCH_FAST, "match: unknown prec_mode %d", prec_mode);
    VFM_CHECK_ARG(d % 128 == 0 && d >= 128 && d <= 768, "match(FAST): d must be in {128,256,384,512,640,768}, got %d", d);
    unsigned char* p = static_cast<unsigned char*>(ws);
    void* qprep = p;
    void* bprep = p + vfm_match_prepared_bytes(n, d);
    void* sws = p + vfm_match_prepared_bytes(n, d) + vfm_match_prepared_bytes(m, d);
    VFM_CHECK_ARG(m < (1ll << 31) - 256 && n < (1ll << 31) - 256, "match: more than 2^31 rows");
    int rc = do_prepare2(b, m, bprep, q, n, qprep, d, st, !use_i8(d, n, m, gated));
    if (rc) return rc;
    // one-shot calls have no feedback loop: packed top-2 records, the robust kind (real lifted descriptors are duplicate-rich)
    rc = do_search_coarse(qprep, n, bprep, m, d, sws, st, false, true, gated, VFM_RECORDS_TOP2);
    if (rc) return rc;
    return do_search_finish(q, qprep, n, b, bprep, m, d, idx_out, sim_out, sws, st, gated, gated ? gate : -__builtin_inff(),
                            VFM_RECORDS_TOP2);
}

VFM_EXPORT int vfm_match_ip_top1(const float* q, int64_t n, const float* b, int64_t m, int d, int prec_mode,
                                 int64_t* idx_out, float* sim_out, void* ws, size_t ws_bytes, vfm_stream_t stream) {
    return ip_top1(q, n, b, m, d, prec_mode, false, 0.0f, idx_out, sim_out, ws, ws_bytes, stream);
}
VFM_EXPORT int vfm_match_ip_top1_gated(const float* q, int64_t n, const float* b, int64_t m, int d, int prec_mode, float gate,
                                       int64_t* idx_out, float* sim_out, void* ws, size_t ws_bytes, vfm_stream_t stream) {
    VFM_CHECK_ARG(gate == gate, "match: gate is NaN");
    return ip_top1(q, n, b, m, d, prec_mode, true, gate, idx_out, sim_out, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------
// profiling hooks: HIP events around the dominant kernel (match_coarse_kernel), on the stream the
// kernel is launched on.  Used by bench.py for roofline.achieved.
// ---------------------------------------------------------------------------------------------
// A/B switch for the coarse kernel variant (1 = 8 waves x 32 queries, 2 = 4 waves x 64 queries, 0 = default)
// Counters of the last search that used workspace `ws` (sizes as passed to that search): out64_host[0] queries
// decided by the all-pairs fallback, [1] queries refined in fp32, [2] candidate entries after select, [3] rows kept
// by the refinement, [8 + b] queries with 2^(b-1) < entries <= 2^b.  Synchronises the device.
VFM_EXPORT int vfm_debug_match_stats(void* ws, int64_t n, int64_t m, int32_t* out64_host) {
    VFM_CHECK_ARG(ws && out64_host, "match_stats: bad arguments");
    SearchWs w = carve_search(ws, n, m);
    VFM_CHECK_HIP(hipDeviceSynchronize());
    VFM_CHECK_HIP(hipMemcpy(out64_host, w.fb_count, 64 * sizeof(int), hipMemcpyDeviceToHost));
    return VFM_OK;
}

// The int8 image of a prepared operand, unpacked on the host: q8_host[rows][d] (int8), step_host[rows] (the row's group
// step), err_host[rows] (E), gerr_host[rows] (its group's maximum E).  Tests only (the bound of prep_chunk_kernel is checked
// pair by pair against fp64 scores).  Synchronises the device.
VFM_EXPORT int vfm_debug_i8_rows(const void* prepared, int64_t rows, int d, int8_t* q8_host, float* step_host, float* err_host,
                                 float* gerr_host) {
    VFM_CHECK_ARG(prepared && rows > 0 && i8_capable(d) && q8_host && step_host && err_host && gerr_host, "i8_rows: bad arguments");
    Prepared p = carve_prepared(const_cast<void*>(prepared), rows, d);
    const int64_t rp = rows_padded(rows);
    const size_t units = (size_t)rp / TILE_ROWS * (size_t)(d / 32) * 64;
    std::vector<uint4> tiles(units);
    std::vector<float> gstep((size_t)rp / I8_GROUP), gerr((size_t)rp / I8_GROUP);
    VFM_CHECK_HIP(hipDeviceSynchronize());
    VFM_CHECK_HIP(hipMemcpy(tiles.data(), p.tiles8, units * sizeof(uint4), hipMemcpyDeviceToHost));
    VFM_CHECK_HIP(hipMemcpy(gstep.data(), p.gstep, gstep.size() * sizeof(float), hipMemcpyDeviceToHost));
    VFM_CHECK_HIP(hipMemcpy(gerr.data(), p.gerr, gerr.size() * sizeof(float), hipMemcpyDeviceToHost));
    VFM_CHECK_HIP(hipMemcpy(err_host, p.err, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost));
    const int upt = (d / 32) * 64;  // units per tile
    for (int64_t r = 0; r < rows; ++r) {
        const int64_t tile = r / TILE_ROWS, pp = r % TILE_ROWS;
        for (int u = 0; u < d / 16; ++u) {  // unit u = 2 s + h holds k = 16 u .. 16 u + 15
            const int8_t* src = reinterpret_cast<const int8_t*>(&tiles[(size_t)tile * upt + (size_t)u * 32 + pp]);
            for (int k = 0; k < 16; ++k) q8_host[r * (int64_t)d + 16 * u + k] = src[k];
        }
        step_host[r] = gstep[(size_t)(r / I8_GROUP)];
        gerr_host[r] = gerr[(size_t)(r / I8_GROUP)];
    }
    return VFM_OK;
}

// The fp6 image of a prepared operand (VFM_PREPARE_MX6), dequantised on the host: v6_host[rows][d] (float: code value x block
// scale), err_host[rows] (E of the fp6 image, slack included), gerr_host[rows] (its group's maximum).  Tests only.
VFM_EXPORT int vfm_debug_mx6_rows(const void* prepared, int64_t rows, int d, float* v6_host, float* err_host, float* gerr_host) {
    VFM_CHECK_ARG(prepared && rows > 0 && mx6_half_width(d) && v6_host && err_host && gerr_host, "mx6_rows: bad arguments");
    Prepared p = carve_prepared(const_cast<void*>(prepared), rows, d);
    const int64_t rp = rows_padded(rows);
    const int ks = d / 64;
    const size_t tb = (size_t)mx6_tile_bytes(ks);
    const size_t units = (size_t)rp / TILE_ROWS * (tb / 16);
    std::vector<uint4> tiles(units);
    std::vector<float> gerr((size_t)rp / I8_GROUP);
    VFM_CHECK_HIP(hipDeviceSynchronize());
    VFM_CHECK_HIP(hipMemcpy(tiles.data(), p.tiles6, units * sizeof(uint4), hipMemcpyDeviceToHost));
    VFM_CHECK_HIP(hipMemcpy(gerr.data(), p.gerr6, gerr.size() * sizeof(float), hipMemcpyDeviceToHost));
    VFM_CHECK_HIP(hipMemcpy(err_host, p.err6, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost));
    const unsigned char* img = reinterpret_cast<const unsigned char*>(tiles.data());
    for (int64_t r = 0; r < rows; ++r) {
        const int64_t tile = r / TILE_ROWS, pp = r % TILE_ROWS;
        for (int blk = 0; blk < d / 32; ++blk) {   // block = (k-step s, half h): MFMA lane h * 32 + pp, columns 64 s + 32 h ...
            const int s = blk >> 1, h = blk & 1, l6 = h * 32 + (int)pp;
            unsigned char op[32] = {0};
            const unsigned char* lo = img + (size_t)tile * tb + mx6_code_a(s, l6);   // code bytes 0 .. 15
            const unsigned char* hi = img + (size_t)tile * tb + mx6_code_b(s, l6);   // code bytes 16 .. 23
            for (int i = 0; i < 16; ++i) op[i] = lo[i];
            for (int i = 0; i < 8; ++i) op[16 + i] = hi[i];
            const float scale = ldexpf(1.0f, (int)img[(size_t)tile * tb + mx6_scale_at(ks, s, l6)] - 127);
            for (int f = 0; f < 32; ++f) {
                const int bit = 6 * f;
                const unsigned w = (unsigned)op[bit >> 3] | ((unsigned)op[(bit >> 3) + 1] << 8);
                const unsigned code = (w >> (bit & 7)) & 63u;
                const int e = (code >> 3) & 3, mnt = code & 7;
                float v = e == 0 ? mnt / 8.0f : (1.0f + mnt / 8.0f) * (float)(1 << (e - 1));
                if (code & 32u) v = -v;
                v6_host[r * (int64_t)d + 32 * blk + f] = v * scale;
            }
        }
        gerr_host[r] = gerr[(size_t)(r / I8_GROUP)];
    }
    return VFM_OK;
}

// E of the fp6 image over the first d / 2 columns (what the half-width fp6 kinds bound with) and its group maximum.  Tests only.
VFM_EXPORT int vfm_debug_mx6_half_err(const void* prepared, int64_t rows, int d, float* errh_host, float* gerrh_host) {
    VFM_CHECK_ARG(prepared && rows > 0 && mx6_half_width(d) && errh_host && gerrh_host, "mx6_half_err: bad arguments");
    Prepared p = carve_prepared(const_cast<void*>(prepared), rows, d);
    std::vector<float> g((size_t)rows_padded(rows) / I8_GROUP);
    VFM_CHECK_HIP(hipDeviceSynchronize());
    VFM_CHECK_HIP(hipMemcpy(errh_host, p.err6h, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost));
    VFM_CHECK_HIP(hipMemcpy(g.data(), p.gerr6h, g.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < rows; ++r) gerrh_host[r] = g[(size_t)(r / I8_GROUP)];
    return VFM_OK;
}





VFM_EXPORT int vfm_prof_events_create(void** start, void** stop) {
    VFM_CHECK_ARG(start && stop, "prof: null pointer");
    hipEvent_t a, b;
    VFM_CHECK_HIP(hipEventCreate(&a));
    VFM_CHECK_HIP(hipEventCreate(&b));
    *start = a;
    *stop = b;
    return VFM_OK;
}
VFM_EXPORT int vfm_prof_arm(void* start, void* stop) {
    g_prof_start = (hipEvent_t)start;
    g_prof_stop = (hipEvent_t)stop;
    return VFM_OK;
}
VFM_EXPORT int vfm_prof_elapsed_ms(void* start, void* stop, float* ms_host) {
    VFM_CHECK_ARG(start && stop && ms_host, "prof: null pointer");
    VFM_CHECK_HIP(hipEventSynchronize((hipEvent_t)stop));
    VFM_CHECK_HIP(hipEventElapsedTime(ms_host, (hipEvent_t)start, (hipEvent_t)stop));
    return VFM_OK;
}
VFM_EXPORT int vfm_prof_events_destroy(void* start, void* stop) {
    if (start) (void)hipEventDestroy((hipEvent_t)start);
    if (stop) (void)hipEventDestroy((hipEvent_t)stop);
    return VFM_OK;
}

