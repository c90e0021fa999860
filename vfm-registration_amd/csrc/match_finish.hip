// match_finish.hip -- from coarse records to the oracle's answer: candidate selection
// (match_select_kernel), int8 rescan of candidate chunks (match_rescan_kernel), fp32 refinement of crowded lists
// (match_refine_kernel, match_filter_refine_kernel), the exact fp64 decision (match_rescore_kernel), the all-pairs fp64
// kernel (EXACT mode and overflow fallback) and the cosine threshold + compaction (VoxelHashMap.cpp:501-511, 587-600).
#include "match_internal.h"

namespace vfmm {
namespace {

// ---------------------------------------------------------------------------------------------
// selection: coarse max per query and the candidate chunks inside the error window
// cand entry: (chunk << 8) | (rescan << 7) | local row
// ---------------------------------------------------------------------------------------------
constexpr int SELECT_GROUPS = 16;  // waves per 64 queries: enough loads in flight to saturate HBM on the sweep
__global__ __launch_bounds__(64 * SELECT_GROUPS) void match_select_kernel(const uint2* __restrict__ partials, int nchunks, int npad,
                                                           int64_t n, int first_pad_chunk,
                                                           const unsigned* __restrict__ qmax,
                                                           const float* __restrict__ invq, float window,
                                                           I8Bounds ib, float gate, int chunk_lds, int* __restrict__ cand_cnt,
                                                           unsigned* __restrict__ cand, int cap, int* __restrict__ fb_count,
                                                           int* __restrict__ fb_list, int stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char select_smem[];  // int8 records: the chunks' (step, max E)
    __shared__ int lcnt[64];
    __shared__ int lresc[64];     // int8 top-2 records: whole-chunk entries among them
    __shared__ unsigned lub[64];  // int8 records: float_key of the largest upper bound over the query's chunks
    const int qq = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 64 + qq;
    if (g == 0) {
        lcnt[qq] = 0;
        lresc[qq] = 0;
        lub[qq] = 0u;
    }
    float2* lchunk = reinterpret_cast<float2*>(select_smem);
    if (ib.qerr && chunk_lds)  // every thread walks ~nchunks / 16 chunks: their (step, max E) once per workgroup into LDS
        for (int c = threadIdx.x; c < nchunks; c += 64 * SELECT_GROUPS) lchunk[c] = make_float2(ib.bstep[c], ib.berr[c]);
    __syncthreads();
    if (ib.qerr) {
        // Records of the int8 pass: integer scores in the units of (query group step) x (map chunk step).  In exact score
        // units, with A = (1 + 2^-13) E_q and B_c = (1 + 2^-13 + E_q) max E of chunk c (prep_chunk_kernel):
        //     lower_c = s_q s_c S(c) - A - B_c  <=  best exact score of chunk c  <=  s_q s_c S(c) + A + B_c = upper_c
        // (S(c): the chunk's best integer score, exact).
        // qlow = max over the un-padded chunks of lower_c, a lower bound of the query's exact maximum, comes from the coarse
        // kernel (qmax holds its float_key).  Every chunk with upper_c >= qlow is a candidate:
        // the oracle's arg-max row is inside one of them, and match_refine_kernel finds the rows (int8 rescan, then fp32).
        // fp32 evaluation of the bounds: three roundings on magnitudes <= 2 -- 1e-6 of slack covers them.
        const float eq = ib.qerr[q], sq = ib.qstep[q >> 7];
        const float A = eq * 1.0001220703125f, mult = 1.0001220703125f + eq, slack = 1.0e-6f;
        const float qlow = key_float(qmax[q]);  // -Inf: no un-padded chunk exists -> every chunk is a candidate
        float maxup = -__builtin_inff();
        if (ib.top2) {
            // packed top-2 records (uint2 [chunk][npad], as the fp16 pass writes them): the best row's index rides in the low
            // 7 bits, so a candidate chunk whose SECOND-best score cannot reach qlow is a single-row entry and needs no rescan
            for (int cb = g; cb < nchunks; cb += 8 * SELECT_GROUPS) {
                uint2 rec[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = cb + SELECT_GROUPS * u;
                    rec[u] = (c < nchunks) ? partials[((size_t)(q >> 5) * nchunks + c) * 32 + (q & 31)] : make_uint2(0u, 0u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = cb + SELECT_GROUPS * u;
                    if (c >= nchunks) continue;
                    const float2 cb2 = chunk_lds ? lchunk[c] : make_float2(ib.bstep[c], ib.berr[c]);
                    const float sc = sq * cb2.x, bound = A + mult * cb2.y + slack;
                    const float up1 = sc * (float)((int)(rec[u].x | 127u) - I8_OFFSET) + bound;
                    maxup = fmaxf(maxup, up1);
                    if (up1 >= qlow) {
                        const int slot = atomicAdd(&lcnt[qq], 1);
                        const float up2 = sc * (float)((int)(rec[u].y | 63u) - I8_OFFSET) + bound;
                        const unsigned rescan = (up2 >= qlow || c >= first_pad_chunk) ? 1u : 0u;
                        if (rescan) atomicAdd(&lresc[qq], 1);
                        if (slot < cap && q < n) cand[(size_t)q * cap + slot] = ((unsigned)c << 8) | (rescan << 7) | (rec[u].x & 127u);
                    }
                }
            }
        } else {
        // [query tile][chunk][32]: best integer score (+ 2^30) of the chunk for the tile's 32 queries
        const unsigned* best = reinterpret_cast<const unsigned*>(partials) + (size_t)(q >> 5) * nchunks * 32 + (q & 31);
        for (int cb = g; cb < nchunks; cb += 8 * SELECT_GROUPS) {
            unsigned rec[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = cb + SELECT_GROUPS * u;
                rec[u] = (c < nchunks) ? best[(size_t)c * 32] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = cb + SELECT_GROUPS * u;
                if (c >= nchunks) continue;
                const float2 cb2 = chunk_lds ? lchunk[c] : make_float2(ib.bstep[c], ib.berr[c]);
                const float sc = sq * cb2.x, bound = A + mult * cb2.y + slack;
                const float up1 = sc * (float)((int)rec[u] - I8_OFFSET) + bound;
                maxup = fmaxf(maxup, up1);
                if (up1 >= qlow) {  // (zero-padded rows score exactly 0: a padded chunk is a candidate only if 0 is inside the window)
                    const int slot = atomicAdd(&lcnt[qq], 1);
                    // (the records hold values only: which rows of the chunk reach qlow is found by match_refine_kernel's
                    // int8 rescan -- every entry is a whole-chunk entry)
                    if (slot < cap && q < n) cand[(size_t)q * cap + slot] = ((unsigned)c << 8) | 128u;
                }
            }
        }
        }
        atomicMax(&lub[qq], float_key(maxup));
    } else {
    // qmax = best coarse score over the un-padded chunks (value bits; accumulated by the coarse
    // kernel).  Chunks >= first_pad_chunk contain zero-padded map rows whose coarse score (exactly
    // 2.0) is meaningless: they do not take part in the maximum and are always rescanned exactly.
    const unsigned m = qmax[q];
    // m == 0: no un-padded chunk exists (map smaller than one chunk) -> every chunk is a candidate
    const float thr_f = __uint_as_float(m) - window;
    const unsigned thr = (m == 0u) ? 0u : (__float_as_uint(thr_f) & ~127u);
    // HBM-bound sweep over this query's records: 8 independent loads in flight per thread
    for (int cb = g; cb < nchunks; cb += 8 * SELECT_GROUPS) {
        uint2 rec[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = cb + SELECT_GROUPS * u;
            rec[u] = (c < nchunks) ? partials[(size_t)c * npad + q] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = cb + SELECT_GROUPS * u;
            if (c < nchunks && (rec[u].x | 127u) >= thr) {
                const int slot = atomicAdd(&lcnt[qq], 1);
                if (slot < cap && q < n) {  // sparse: straight to the query's global list
                    const unsigned rescan = (((rec[u].y | 63u) >= thr) || c >= first_pad_chunk) ? 1u : 0u;
                    cand[(size_t)q * cap + slot] = ((unsigned)c << 8) | (rescan << 7) | (rec[u].x & 127u);
                }
            }
        }
    }
    }
    __syncthreads();
    if (g == 0 && ib.qerr) {
        // load figure of the int8 pass: candidate chunks that match_refine_kernel will rescan (the caller's feedback for
        // choosing between this pass and the fp16 one on duplicate-rich maps: vfm_match_search_rescans_async); one atomic
        // per workgroup
        int mine = 0;
        // (top-2 records: whole-chunk entries count 1, single-row entries 1/32 -- 48 KB of int8 tiles against 1.5 KB of fp32 row)
        if (q < n && invq[q] != 0.0f && !(key_float(lub[qq]) < gate) && lcnt[qq] <= cap)
            mine = ib.top2 ? lresc[qq] + ((lcnt[qq] - lresc[qq]) >> 5) : lcnt[qq];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
        if (qq == 0 && mine > 0) atomicAdd(fb_count + 5, mine);
    }
    if (g == 0 && q < n) {
        const int cnt = lcnt[qq];
        // statistics (vfm_debug_match_stats): [2] candidate entries, [8 + b] queries with 2^(b-1) < entries <= 2^b
        if (stats && invq[q] != 0.0f) {
            atomicAdd(fb_count + 2, cnt);
            int bin = 0;
            while ((1 << bin) < cnt && bin < 15) ++bin;
            atomicAdd(fb_count + 8 + bin, 1);
        }
        if (invq[q] == 0.0f) {
            cand_cnt[q] = 0;  // zero query row: decided directly (index 0, score 0)
        } else if (ib.qerr && key_float(lub[qq]) < gate) {
            // no row of the map can reach the caller's similarity gate (VoxelHashMap.cpp:501-511 drops such queries):
            // the query is not resolved further -- match_rescore_kernel reports (index -1, similarity -2)
            cand_cnt[q] = -2;
        } else if (cnt > cap) {
            cand_cnt[q] = -1;  // overflow: decided by the exact all-pairs kernel
            const int slot = atomicAdd(fb_count, 1);
            fb_list[slot] = (int)q;
        } else {
            cand_cnt[q] = cnt;
        }
    }
}

// Packed top-2 records of the int8 pass ([query tile][chunk][32], 8 bytes each): the int8 / top-2 branch of
// match_select_kernel laid out for the sweep, as match_select_best_kernel below.  A lane owns two consecutive queries of the
// tile and one chunk of a block of four: a wave's load instruction covers 4 chunks x 32 queries = 1 KiB of consecutive
// record bytes.  Queries that provably end below the gate record nothing (see match_select_best_kernel).
constexpr int SELECT_TOP2_WAVES = 8;
__global__ __launch_bounds__(64 * SELECT_TOP2_WAVES) void match_select_top2_kernel(
    const uint2* __restrict__ recs, int nchunks, int64_t n, int first_pad_chunk, const unsigned* __restrict__ qmax,
    const float* __restrict__ invq, I8Bounds ib, float gate, int chunk_lds, int* __restrict__ cand_cnt, unsigned* __restrict__ cand,
    int cap, int* __restrict__ fb_count, int* __restrict__ fb_list, int stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char select_smem[];  // the chunks' (step, max E)
    __shared__ int lcnt[32];
    __shared__ int lresc[32];     // whole-chunk entries among them
    __shared__ unsigned lub[32];  // float_key of the largest upper bound over the query's chunks
    __shared__ unsigned lmaxe;
    __shared__ int ldead[32];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int qt = blockIdx.x;
    if (threadIdx.x < 32) {
        lcnt[threadIdx.x] = 0;
        lresc[threadIdx.x] = 0;
        lub[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) lmaxe = 0u;
    __syncthreads();
    float2* lchunk = reinterpret_cast<float2*>(select_smem);
    if (chunk_lds) {
        float me = 0.0f;
        for (int c = threadIdx.x; c < nchunks; c += 64 * SELECT_TOP2_WAVES) {
            const float2 v = make_float2(ib.bstep[c], ib.berr[c]);
            lchunk[c] = v;
            me = fmaxf(me, v.y);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) me = fmaxf(me, __shfl_xor(me, off));
        if (lane == 0) atomicMax(&lmaxe, float_key(me));
    }
    __syncthreads();
    const int lq = (lane & 15) * 2, lc = lane >> 4;
    const int64_t q0 = (int64_t)qt * 32 + lq;   // the lane's two queries
    const float sq = ib.qstep[q0 >> 7], slack = 1.0e-6f;
    float A[2], mult[2], qlow[2], maxup[2];
    unsigned deadmask = 0u;
    const float maxe = chunk_lds ? key_float(lmaxe) : __builtin_inff();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float eq = ib.qerr[q0 + j];
        A[j] = eq * 1.0001220703125f;
        mult[j] = 1.0001220703125f + eq;
        qlow[j] = key_float(qmax[q0 + j]);
        maxup[j] = -__builtin_inff();
        const bool dead = qlow[j] + 2.0f * (A[j] + mult[j] * maxe) + 1.0e-5f < gate;
        deadmask |= dead ? (1u << j) : 0u;
    }
    const uint4* src = reinterpret_cast<const uint4*>(recs + (size_t)qt * nchunks * 32) + (lane & 15);
    for (int c = first_pad_chunk; c < nchunks && deadmask; ++c) {  // padded chunks are not covered by qlow
        const uint4 r = src[(size_t)c * 16];
        const unsigned r1[2] = {r.x, r.z};
        const float sc = sq * ib.bstep[c], be = ib.berr[c];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (!(sc * (float)((int)(r1[j] | 127u) - I8_OFFSET) + (A[j] + mult[j] * be + slack) < gate)) deadmask &= ~(1u << j);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (q0 + j >= n || invq[q0 + j] == 0.0f) deadmask |= 1u << j;
        if (threadIdx.x < 16) ldead[lq + j] = (deadmask >> j) & 1u;
    }
    const int nblocks = (nchunks + 3) >> 2;
    for (int cb0 = wave; cb0 < nblocks; cb0 += 8 * SELECT_TOP2_WAVES) {
        uint4 rec[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cb = cb0 + SELECT_TOP2_WAVES * u;
            rec[u] = (cb * 4 + lc < nchunks) ? src[((size_t)cb * 4 + lc) * 16] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = (cb0 + SELECT_TOP2_WAVES * u) * 4 + lc;
            if (c >= nchunks) continue;
            const float2 cb2 = chunk_lds ? lchunk[c] : make_float2(ib.bstep[c], ib.berr[c]);
            const float sc = sq * cb2.x;
            const unsigned r1[2] = {rec[u].x, rec[u].z}, r2[2] = {rec[u].y, rec[u].w};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bound = A[j] + mult[j] * cb2.y + slack;
                const float up1 = sc * (float)((int)(r1[j] | 127u) - I8_OFFSET) + bound;
                maxup[j] = fmaxf(maxup[j], up1);
                if (up1 >= qlow[j] && !((deadmask >> j) & 1u)) {
                    const int slot = atomicAdd(&lcnt[lq + j], 1);
                    // the best row's index rides in the low 7 bits: a candidate chunk whose SECOND-best score cannot reach
                    // qlow is a single-row entry and needs no rescan
                    const float up2 = sc * (float)((int)(r2[j] | 63u) - I8_OFFSET) + bound;
                    const unsigned rescan = (up2 >= qlow[j] || c >= first_pad_chunk) ? 1u : 0u;
                    if (rescan) atomicAdd(&lresc[lq + j], 1);
                    if (slot < cap) cand[(size_t)(q0 + j) * cap + slot] = ((unsigned)c << 8) | (rescan << 7) | (r1[j] & 127u);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v = maxup[j];
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
        if (lane < 16) atomicMax(&lub[lq + j], float_key(v));
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // one wave: the tile's 32 queries
        const int qq = lane & 31;
        const int64_t q = (int64_t)qt * 32 + qq;
        const bool live = lane < 32 && q < n;
        const int cnt = lcnt[qq];
        // load figure (vfm_match_search_rescans_async): whole-chunk entries count 1, single-row entries 1/32 -- 48 KB of int8
        // tiles against 1.5 KB of fp32 row
        int mine = 0;
        if (live && invq[q] != 0.0f && !(key_float(lub[qq]) < gate) && cnt <= cap) mine = lresc[qq] + ((cnt - lresc[qq]) >> 5);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0 && mine > 0) atomicAdd(fb_count + 5, mine);
        if (live) {
            if (stats && invq[q] != 0.0f) {
                atomicAdd(fb_count + 2, cnt);
                int bin = 0;
                while ((1 << bin) < cnt && bin < 15) ++bin;
                atomicAdd(fb_count + 8 + bin, 1);
            }
            const bool dead = ldead[qq] != 0;
            if (invq[q] == 0.0f) {
                cand_cnt[q] = 0;  // zero query row: decided directly (index 0, score 0)
            } else if (key_float(lub[qq]) < gate) {
                cand_cnt[q] = -2;  // no row of the map can reach the caller's similarity gate
            } else if (cnt > cap || dead) {
                cand_cnt[q] = -1;  // overflow (or a "dead" query lifted past the gate after all): the all-pairs kernel decides
                const int slot = atomicAdd(fb_count, 1);
                fb_list[slot] = (int)q;
            } else {
                cand_cnt[q] = cnt;
            }
        }
    }
}

// Half-width pass (VFM_RECORDS_HALF): the records hold the best integer score of every (query, chunk) over the FIRST d / 2
// columns.  With r_q = |second half of the normalised query|, R_c = max over the chunk's rows of |second half of the row|
// (both rounded up, prep_chunk_kernel) and the quantisation bound A + B_c of the full rows (an upper bound for any subset of
// columns):      best exact score of chunk c  <=  s_q s_c S_half(c) + A + B_c + r_q R_c       (Cauchy-Schwarz on the other half)
// A chunk survives for a query iff that bound reaches the gate; the survivors are binned per chunk (or left in the query's
// list) for the full-width int8 rescan, whose hit test is then the gate itself.  A query without survivors provably has no
// match (cand_cnt = -2).  Same sweep layout as match_select_best_kernel.
__global__ __launch_bounds__(64 * 8) void match_select_half_kernel(
    const unsigned* __restrict__ best, int nchunks, int64_t n, const float* __restrict__ invq, I8Bounds ib,
    const float* __restrict__ qrest, const float* __restrict__ grest, float gate, int chunk_lds, int* __restrict__ cand_cnt,
    unsigned* __restrict__ cand, int cap, int* __restrict__ fb_count, int* __restrict__ fb_list, int stats,
    unsigned* __restrict__ bin_cnt, int* __restrict__ bins, int bin_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char select_smem[];  // the chunks' (step, max E) and max |rest|
    __shared__ int lcnt[32];
    __shared__ int lov[32];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int qt = blockIdx.x;
    if (threadIdx.x < 32) {
        lcnt[threadIdx.x] = 0;
        lov[threadIdx.x] = 0;
    }
    float2* lchunk = reinterpret_cast<float2*>(select_smem);
    float* lrest = reinterpret_cast<float*>(lchunk + nchunks);
    if (chunk_lds)
        for (int c = threadIdx.x; c < nchunks; c += 64 * 8) {
            lchunk[c] = make_float2(ib.bstep[c], ib.berr[c]);
            lrest[c] = grest[c];
        }
    __syncthreads();
    const int lq = (lane & 7) * 4, lc = lane >> 3;
    const int64_t q0 = (int64_t)qt * 32 + lq;   // the lane's four queries (rows of the padded tile always exist)
    const float sq = ib.qstep[q0 >> 7], slack = 1.0e-6f;
    float A[4], mult[4], rq[4];
    unsigned livemask = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float eq = ib.qerr[q0 + j];
        A[j] = eq * 1.0001220703125f;
        mult[j] = 1.0001220703125f + eq;
        rq[j] = qrest[q0 + j];
        if (q0 + j < n && invq[q0 + j] != 0.0f) livemask |= 1u << j;
    }
    const uint4* src = reinterpret_cast<const uint4*>(best + (size_t)qt * nchunks * 32) + lane;
    const int nblocks = (nchunks + 7) >> 3;
    for (int cb0 = wave; cb0 < nblocks; cb0 += 8 * 8) {
        if (cand) {
            // The guard, early: once four queries of a lane average 4 x HALF_GUARD_PER_QUERY surviving chunks the search is going to
            // take the gate pass anyway (half_guard_kernel) -- raise its flag now, leave a load figure that says so, and stop: on
            // descriptors that are all alike the full sweep (31 million survivors, each a few atomics) took 21 ms.  Every workgroup
            // stops at the flag; the gate pass ignores what the lists hold.
            if (__hip_atomic_load(fb_count + HALF_GUARD_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        }
        bool stop = false;
        uint4 rec[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cb = cb0 + 8 * u;
            rec[u] = (cb * 8 + lc < nchunks) ? src[(size_t)cb * 64] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = (cb0 + 8 * u) * 8 + lc;
            if (c >= nchunks || stop) continue;
            if (cand && lcnt[lq] + lcnt[lq + 1] + lcnt[lq + 2] + lcnt[lq + 3] > 16 * HALF_GUARD_PER_QUERY) {
                if (atomicExch(fb_count + HALF_GUARD_FLAG, 1) == 0) atomicExch(fb_count + 5, 0x3FFFFFFF);
                stop = true;
                continue;
            }
            const float2 cb2 = chunk_lds ? lchunk[c] : make_float2(ib.bstep[c], ib.berr[c]);
            const float rb = chunk_lds ? lrest[c] : grest[c];
            const float sc = sq * cb2.x;
            const unsigned r4[4] = {rec[u].x, rec[u].y, rec[u].z, rec[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // fp32 evaluation: five roundings on magnitudes <= 2 -- 2e-6 of slack covers them
                const float up = sc * (float)((int)r4[j] - I8_OFFSET) + (A[j] + mult[j] * cb2.y + slack) + (rq[j] * rb + slack);
                if (!(up < gate) && ((livemask >> j) & 1u)) {
                    int slot = atomicAdd(&lcnt[lq + j], 1);
                    if (!cand) continue;  // probe (vfm_match_search_probe_half): the survivors are only counted
                    if (bins) {
                        // (a full bin is not touched again: on descriptors that are all alike every query survives in every
                        // chunk, and 31 million atomics on 1563 addresses were most of this kernel's time there)
                        const unsigned seen = __hip_atomic_load(&bin_cnt[(size_t)c * BIN_CNT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned pos = seen >= (unsigned)bin_cap ? seen : atomicAdd(&bin_cnt[(size_t)c * BIN_CNT_STRIDE], 1u);
                        if (pos < (unsigned)bin_cap) {
                            bins[(size_t)c * bin_cap + pos] = (int)(q0 + j);
                            slot = cap;
                        } else {
                            slot = atomicAdd(&lov[lq + j], 1);
                        }
                    }
                    if (slot < cap) cand[(size_t)(q0 + j) * cap + slot] = ((unsigned)c << 8) | 128u;  // whole-chunk entry
                }
            }
        }
        if (stop) break;
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // one wave: the tile's 32 queries
        const int qq = lane & 31;
        const int64_t q = (int64_t)qt * 32 + qq;
        const bool live = lane < 32 && q < n;
        const int cnt = lcnt[qq];
        int mine = (live && (cnt <= cap || !cand)) ? cnt : 0;   // load figure (vfm_match_search_rescans_async): surviving chunks
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0 && mine > 0) atomicAdd(fb_count + 5, mine);
        if (live && cand) {
            if (stats && invq[q] != 0.0f) {
                atomicAdd(fb_count + 2, cnt);
                int bin = 0;
                while ((1 << bin) < cnt && bin < 15) ++bin;
                atomicAdd(fb_count + 8 + bin, 1);
            }
            if (invq[q] == 0.0f) {
                cand_cnt[q] = 0;  // zero query row: decided directly (index 0, score 0)
            } else if (cnt == 0) {
                cand_cnt[q] = -2;  // no chunk can hold a row at the gate
            } else if (cnt > cap) {
                cand_cnt[q] = -1;  // more surviving chunks than a list holds: the all-pairs kernel decides
                const int slot = atomicAdd(fb_count, 1);
                fb_list[slot] = (int)q;
            } else {
                cand_cnt[q] = bins ? lov[qq] : cnt;
            }
        }
    }
}

// VFM_RECORDS_MX6_HALF_FUSED: the fused fp6 half-width kernel (match_coarse_mx6.hip) has tested every (query, chunk) pair against
// the gate itself and left each workgroup's survivors in a slot of the record buffer -- [count, query block, first chunk, overflow]
// + entries (chunk of the slice << 9 | query of the block).  This kernel does the placement match_select_half_kernel does for its
// survivors: into the chunk's rescan bin, past a full bin into the query's own list.  One wave per slot; ~11 000 entries at C2 on
// SURVEY D.2 data (0.56 per query) against the 122 MB sweep of the records this replaces.  With the guard up (a slot overflowed:
// descriptors that are all alike) nothing is placed: match_gatepass_kernel decides every query.
__global__ __launch_bounds__(256) void match_bin_survivors_kernel(const unsigned* __restrict__ surv, int slot_words, const int* __restrict__ fb_count,
                                                                  unsigned* __restrict__ bin_cnt, int* __restrict__ bins, int bin_cap,
                                                                  int* __restrict__ cand_cnt, unsigned* __restrict__ cand, int cap) {
    if (fb_count[HALF_GUARD_FLAG] != 0) return;
    const int nslots = fb_count[MX6_GRID_SLOT];
    const int lane = lane_id(), nwaves = (int)gridDim.x * 4;
    for (int sidx = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); sidx < nslots; sidx += nwaves) {
        const unsigned* slot = surv + (size_t)sidx * slot_words;
        const unsigned cnt = slot[0], qb = slot[1], c0 = slot[2];
        // (round 5: a workgroup of the coarse kernel holds 512 queries -- two 32-query tiles per wave, entries = chunk << 9 | query -- or 768 --
        // three tiles, chunk << 10 | query --; it says which in bits 8 .. of the header's fourth word)
        const bool three = cnt > 0u && (slot[3] >> 8) == 3u;
        const unsigned qshift = three ? 10u : 9u, qper = three ? 768u : 512u;
        for (unsigned i = (unsigned)lane; i < cnt; i += 64u) {
            const unsigned e = slot[4 + i];
            const int chunk = (int)(c0 + (e >> qshift));
            const int64_t q = (int64_t)qb * qper + (int64_t)(e & ((1u << qshift) - 1u));
            const unsigned pos = atomicAdd(&bin_cnt[(size_t)chunk * BIN_CNT_STRIDE], 1u);
            if (pos < (unsigned)bin_cap) {
                bins[(size_t)chunk * bin_cap + pos] = (int)q;
            } else {   // a full bin leaves the entry in the query's own list (match_rescan_kernel)
                const int own = atomicAdd(&cand_cnt[q], 1);
                if (own < cap) cand[(size_t)q * cap + own] = ((unsigned)chunk << 8) | 128u;
            }
        }
    }
}

// Best-score records of the int8 pass ([query tile][chunk][32], 4 bytes each): the same decision as the int8 branch of
// match_select_kernel, laid out for the sweep.  One workgroup per query tile; a lane owns four consecutive queries of the
// tile and one chunk of a block of eight, so a wave's load instruction covers 8 chunks x 32 queries = 1 KiB of consecutive
// record bytes (the general kernel reads two 128-byte pieces per instruction: 103 us for the 125 MB of C2's records; this
// one is bound by the sweep itself).
constexpr int SELECT_BEST_WAVES = 8;
constexpr int SELECT_BEST_LCAND = 6144;   // candidates of a query tile staged in the LDS (192 per query; beyond: the direct path)
__global__ __launch_bounds__(64 * SELECT_BEST_WAVES) void match_select_best_kernel(
    const unsigned* __restrict__ best, int nchunks, int64_t n, const unsigned* __restrict__ qmax, const float* __restrict__ invq,
    I8Bounds ib, float gate, int chunk_lds, int* __restrict__ cand_cnt, unsigned* __restrict__ cand, int cap,
    int* __restrict__ fb_count, int* __restrict__ fb_list, int stats, unsigned* __restrict__ bin_cnt, int* __restrict__ bins,
    int first_pad_chunk, int bin_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char select_smem[];  // the chunks' (step, max E)
    __shared__ int lcnt[32];      // candidate chunks of the query
    __shared__ int lov[32];       // bins != NULL: those of them that did not fit their chunk's bin (they go to the query's own list)
    __shared__ unsigned lub[32];  // float_key of the largest upper bound over the query's chunks
    __shared__ unsigned lmaxe;    // float_key of the largest max E over the chunks
    __shared__ int ldead[32];     // the query records nothing (see below)
    // The tile's candidates, (query of the tile << 27) | chunk: a candidate costs two LDS atomics inside the sweep; the global
    // side -- the chunk's bin (a returning atomic on one of ~1500 addresses) or the query's own list -- is done behind the
    // sweep for all of them at once.  Inside the sweep every candidate had stalled its wave for a round trip: 151 us against
    // 70 at 12 candidate chunks per query (lifted descriptors) and 317 at 46 (the fp6 pass's bounds).
    __shared__ unsigned lcand[SELECT_BEST_LCAND];
    __shared__ int lcand_n;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int qt = blockIdx.x;
    if (threadIdx.x < 32) {
        lcnt[threadIdx.x] = 0;
        lov[threadIdx.x] = 0;
        lub[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) {
        lmaxe = 0u;
        lcand_n = 0;
    }
    __syncthreads();
    float2* lchunk = reinterpret_cast<float2*>(select_smem);
    unsigned* lhist = reinterpret_cast<unsigned*>(lchunk + nchunks);   // chunk_lds == 2: the tile's candidates per chunk, then the bins' positions
    const bool hist = chunk_lds == 2 && bins != nullptr;
    if (chunk_lds) {
        float me = 0.0f;
        for (int c = threadIdx.x; c < nchunks; c += 64 * SELECT_BEST_WAVES) {
            const float2 v = make_float2(ib.bstep[c], ib.berr[c]);
            lchunk[c] = v;
            if (hist) lhist[c] = 0u;
            me = fmaxf(me, v.y);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) me = fmaxf(me, __shfl_xor(me, off));
        if (lane == 0) atomicMax(&lmaxe, float_key(me));
    }
    __syncthreads();
    const int lq = (lane & 7) * 4, lc = lane >> 3;
    const int64_t q0 = (int64_t)qt * 32 + lq;   // the lane's four queries (rows of the padded tile always exist)
    const float sq = ib.qstep[q0 >> 7], slack = 1.0e-6f;
    float A[4], mult[4], qlow[4], maxup[4];
    // "dead": the query provably ends below the gate, so its candidates are not even recorded (at C2 the unmatched half of
    // the scan would write sixteen entries per query).  Every un-padded chunk has upper_c = lower_c + 2 (A + mult E_c) up
    // to roundings of 1e-6 and lower_c <= qlow, so the largest upper bound over them is below qlow + 2 (A + mult max E)
    // + 1e-5; the (at most two) padded chunks are not covered by qlow: their records are looked at here.  Should the sweep
    // find the largest upper bound at or above the gate after all, the query goes to the all-pairs kernel (epilogue).
    // Zero query rows (decided directly) and the rows that pad the last tile record nothing either.
    unsigned deadmask = 0u;
    const float maxe = chunk_lds ? key_float(lmaxe) : __builtin_inff();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float eq = ib.qerr[q0 + j];
        A[j] = eq * 1.0001220703125f;
        mult[j] = 1.0001220703125f + eq;
        qlow[j] = key_float(qmax[q0 + j]);  // -Inf: no un-padded chunk exists -> every chunk is a candidate
        maxup[j] = -__builtin_inff();
        const bool dead = qlow[j] + 2.0f * (A[j] + mult[j] * maxe) + 1.0e-5f < gate;
        deadmask |= dead ? (1u << j) : 0u;
    }
    for (int c = first_pad_chunk; c < nchunks && deadmask; ++c) {
        const uint4 r = reinterpret_cast<const uint4*>(best + ((size_t)qt * nchunks + c) * 32)[lane & 7];
        const unsigned r4[4] = {r.x, r.y, r.z, r.w};
        const float sc = sq * ib.bstep[c], be = ib.berr[c];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (!(sc * (float)((int)r4[j] - I8_OFFSET) + (A[j] + mult[j] * be + slack) < gate)) deadmask &= ~(1u << j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (q0 + j >= n || invq[q0 + j] == 0.0f) deadmask |= 1u << j;
        if (threadIdx.x < 8) ldead[lq + j] = (deadmask >> j) & 1u;
    }
    const uint4* src = reinterpret_cast<const uint4*>(best + (size_t)qt * nchunks * 32) + lane;
    const int nblocks = (nchunks + 7) >> 3;
    // candidate (query qq of the tile, chunk c): chunk-major rescan -> the query joins the chunk's bin, a full bin leaves the entry
    // with the query; otherwise the entry goes to the query's own list (lov counts the list's entries either way)
    // (crowded: a tile with more than 64 candidate chunks per query looks at the counter first -- on descriptors that are all alike
    // every bin is full after the first tiles, and 31 million atomics on 1563 addresses were most of the kernel's time then; below
    // that the look is a second trip to the memory side per candidate for nothing)
    auto place = [&](int qq, int c, bool crowded) {
        const int64_t q = (int64_t)qt * 32 + qq;
        if (bins) {
            const unsigned seen = crowded ? __hip_atomic_load(&bin_cnt[(size_t)c * BIN_CNT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const unsigned pos = seen >= (unsigned)bin_cap ? seen : atomicAdd(&bin_cnt[(size_t)c * BIN_CNT_STRIDE], 1u);
            if (pos < (unsigned)bin_cap) {
                bins[(size_t)c * bin_cap + pos] = (int)q;
                return;
            }
        }
        const int slot = atomicAdd(&lov[qq], 1);
        if (slot < cap) cand[(size_t)q * cap + slot] = ((unsigned)c << 8) | 128u;  // whole-chunk entry
    };
    for (int cb0 = wave; cb0 < nblocks; cb0 += 8 * SELECT_BEST_WAVES) {
        uint4 rec[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cb = cb0 + SELECT_BEST_WAVES * u;
            rec[u] = (cb * 8 + lc < nchunks) ? src[(size_t)cb * 64] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = (cb0 + SELECT_BEST_WAVES * u) * 8 + lc;
            if (c >= nchunks) continue;
            const float2 cb2 = chunk_lds ? lchunk[c] : make_float2(ib.bstep[c], ib.berr[c]);
            const float sc = sq * cb2.x;
            const unsigned r4[4] = {rec[u].x, rec[u].y, rec[u].z, rec[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bound = A[j] + mult[j] * cb2.y + slack;
                const float up1 = sc * (float)((int)r4[j] - I8_OFFSET) + bound;
                maxup[j] = fmaxf(maxup[j], up1);
                // (zero-padded rows score exactly 0: a padded chunk is a candidate only if 0 is inside the bounds)
                if (up1 >= qlow[j] && !((deadmask >> j) & 1u)) {
                    atomicAdd(&lcnt[lq + j], 1);
                    const int at = atomicAdd(&lcand_n, 1);
                    if (at < SELECT_BEST_LCAND) lcand[at] = ((unsigned)(lq + j) << 27) | (unsigned)c;
                    else place(lq + j, c, true);   // the staging buffer is full: placed on the spot
                }
            }
        }
    }
    __syncthreads();
    {
        const int ncand = lcand_n < SELECT_BEST_LCAND ? lcand_n : SELECT_BEST_LCAND;
        const bool crowded = lcand_n > 32 * 64;
        if (hist) {
            // One atomic per chunk and tile instead of one per candidate: the tile's candidates are counted per chunk in the LDS (a
            // candidate keeps its rank there: at most 32 queries of a tile share a chunk), the chunk's bin is asked once for that many
            // places, and every candidate takes base + rank.  (920 000 returning atomics on 1564 counters were 40 of the kernel's
            // 105 us on lifted descriptors with a common component; what this form leaves depends on how many of a tile's candidates share their chunks.)
            for (int i = threadIdx.x; i < ncand; i += 64 * SELECT_BEST_WAVES) {
                const unsigned e = lcand[i];
                const unsigned rank = atomicAdd(&lhist[e & 0xFFFFu], 1u);
                lcand[i] = e | (rank << 16);
            }
            __syncthreads();
            constexpr int PU = 4;
            for (int c0 = threadIdx.x; c0 < nchunks; c0 += 64 * SELECT_BEST_WAVES * PU) {
                unsigned want[PU], pos[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int c = c0 + 64 * SELECT_BEST_WAVES * u;
                    want[u] = c < nchunks ? lhist[c] : 0u;
                    pos[u] = 0u;
                }
                if (crowded) {
#pragma unroll
                    for (int u = 0; u < PU; ++u)
                        if (want[u]) pos[u] = __hip_atomic_load(&bin_cnt[(size_t)(c0 + 64 * SELECT_BEST_WAVES * u) * BIN_CNT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < PU; ++u)
                    if (want[u] && pos[u] < (unsigned)bin_cap)
                        pos[u] = __hip_atomic_fetch_add(&bin_cnt[(size_t)(c0 + 64 * SELECT_BEST_WAVES * u) * BIN_CNT_STRIDE], want[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int u = 0; u < PU; ++u)
                    if (want[u]) lhist[c0 + 64 * SELECT_BEST_WAVES * u] = pos[u];
            }
            __syncthreads();
            for (int i = threadIdx.x; i < ncand; i += 64 * SELECT_BEST_WAVES) {
                const unsigned e = lcand[i];
                const int qq = (int)(e >> 27), c = (int)(e & 0xFFFFu);
                const unsigned pos = lhist[c] + ((e >> 16) & 31u);
                const int64_t q = (int64_t)qt * 32 + qq;
                if (pos < (unsigned)bin_cap) {
                    bins[(size_t)c * bin_cap + pos] = (int)q;
                } else {
                    const int slot = atomicAdd(&lov[qq], 1);
                    if (slot < cap) cand[(size_t)q * cap + slot] = ((unsigned)c << 8) | 128u;  // whole-chunk entry
                }
            }
        } else
        for (int i = threadIdx.x; i < ncand; i += 64 * SELECT_BEST_WAVES) {
            const unsigned e = lcand[i];
            place((int)(e >> 27), (int)(e & 0x7FFFFFFu), crowded);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = maxup[j];
        v = fmaxf(v, __shfl_xor(v, 8));
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
        if (lane < 8) atomicMax(&lub[lq + j], float_key(v));
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // one wave: the tile's 32 queries
        const int qq = lane & 31;
        const int64_t q = (int64_t)qt * 32 + qq;
        const bool live = lane < 32 && q < n;
        const int cnt = lcnt[qq];
        // load figure of the int8 pass (vfm_match_search_rescans_async): candidate chunks match_rescan_kernel will rescan
        int mine = 0;
        if (live && invq[q] != 0.0f && !(key_float(lub[qq]) < gate) && cnt <= cap) mine = cnt;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0 && mine > 0) atomicAdd(fb_count + 5, mine);
        if (live) {
            if (stats && invq[q] != 0.0f) {
                atomicAdd(fb_count + 2, cnt);
                int bin = 0;
                while ((1 << bin) < cnt && bin < 15) ++bin;
                atomicAdd(fb_count + 8 + bin, 1);
            }
            const bool dead = ldead[qq] != 0;
            if (invq[q] == 0.0f) {
                cand_cnt[q] = 0;  // zero query row: decided directly (index 0, score 0)
            } else if (key_float(lub[qq]) < gate) {
                cand_cnt[q] = -2;  // no row of the map can reach the caller's similarity gate
            } else if (cnt > cap || dead) {
                // overflow, or a padded chunk lifted a query past the gate whose candidates were not recorded: decided by the
                // exact all-pairs kernel
                cand_cnt[q] = -1;
                const int slot = atomicAdd(fb_count, 1);
                fb_list[slot] = (int)q;
            } else {
                // chunk-major rescan: the list holds only the entries that missed their bins; match_rescan_chunk_kernel
                // appends the rows it finds behind what match_rescan_kernel makes of these
                cand_cnt[q] = lov[qq];
            }
        }
    }
}

// int8 pass, chunk-major rescan (best-score records, many queries per chunk): one workgroup per map chunk scores the chunk's
// 128 rows against every query of the chunk's bin.  The query-major kernel below reads 48 KB per (query, chunk) pair --
// 0.5 GB per registration at C2, from the Infinity Cache / HBM because the 77 MB int8 map does not fit the L2s; here the map
// is read once.  Round 3: on the matrix cores.  Wave w holds tile w of the chunk as the MFMA's first operand (its int8
// fragments: 4 KS registers), the bin's queries are taken 32 at a time -- the workgroup gathers their fragment units from the
// scan's int8 tiles into one operand image in the LDS (16 bytes per unit) -- and KS MFMAs per wave give the exact integer
// scores of 32 rows x 32 queries; a row inside the query's bounds is appended to the query's list (one atomic per lane and
// block for all of its hits).  Same integers, same hit test, hence the same rows as the v_dot4 loop it replaces (24 LDS reads +
// 96 dot4 per query and wave: 26 us at C2's 9891 candidates but 0.7 ms at the 220 000 of an ungated Euclidean search).
constexpr int RESCAN_LHITS = 1024;   // rows a workgroup of match_rescan_chunk_kernel stages in the LDS before it touches the lists (1024: 13 KB of static LDS beside the 39 KB ring at d = 384 -- three workgroups per compute unit instead of two)
template <int KS>
constexpr int rescan_ring_depth() { return KS <= 16 ? 3 : 2; }
template <int KS>
constexpr size_t rescan_slot_bytes() { return (size_t)KS * 1024 + 768; }   // operand image + three rows of 64 per-query terms
template <int KS>  // k-steps of 32 columns (d / 32)
__global__ __launch_bounds__(256) void match_rescan_chunk_kernel(int64_t n, int64_t m, I8Bounds ib, const uint4* __restrict__ q8,
                                                                 const uint4* __restrict__ b8, const unsigned* __restrict__ qmax,
                                                                 int* __restrict__ cand_cnt, unsigned* __restrict__ cand, int cap,
                                                                 const unsigned* __restrict__ bin_cnt, const int* __restrict__ bins,
                                                                 int use_gate, float gate, const int* __restrict__ guard, L2Terms l2,
                                                                 int bin_cap, unsigned* __restrict__ hit_cnt, float* __restrict__ cand_up,
                                                                 int slice, const unsigned char* __restrict__ qrows,
                                                                 unsigned* __restrict__ pilot_qmax) {
    // A ring of D slots, one per block of 32 queries: [KS][64] uint4 = the block's queries as ONE MFMA operand image (unit u =
    // (k-step s, half h, column p) comes from query bin[j0 + p]), then [3][64] dwords of per-query terms.  Everything a block needs
    // from global memory arrives by LDS-DMA issued D - 1 blocks ahead (wave w gathers k-steps w, w + 4, ...: its 64 lanes' units
    // are 1 KiB of consecutive LDS), so that the loop holds no compiler-tracked global load -- one would drain the DMA queue at its
    // s_waitcnt -- and a block costs its MFMAs, not its round trips (gather -> barrier -> MFMA -> returning atomic in sequence had
    // cost ~5 us per block: 148 us at 12 candidate chunks per query, 616 at 46).
    extern __shared__ __attribute__((aligned(16))) unsigned char rescan_smem[];
    constexpr int D = rescan_ring_depth<KS>();
    constexpr unsigned SLOT = (unsigned)rescan_slot_bytes<KS>();
    constexpr int TILE_U4 = KS * 64;
    constexpr int NG = KS / 4;      // gathered 1 KiB pieces per wave and block (KS % 4 == 0)
    __shared__ int lbin[RESCAN_SLICE];
    // The workgroup's hits, staged: the global side (a returning atomic on the query's list length, then the entry) is done
    // for all of them at once behind the loop.
    __shared__ int lhq[RESCAN_LHITS];
    __shared__ unsigned char lhr[RESCAN_LHITS];
    __shared__ float lhu[RESCAN_LHITS];   // the row's upper bound (cand_up)
    __shared__ int lhit_n;
    const int c = blockIdx.x;
    if (guard && *guard) return;   // half-width pass, too many survivors: match_gatepass_kernel has decided every query
    const unsigned filled = bin_cnt[(size_t)c * BIN_CNT_STRIDE];
    const int nall = filled < (unsigned)bin_cap ? (int)filled : bin_cap;
    const int jbeg = blockIdx.y * slice;   // a long bin is shared by the workgroups (c, 0), (c, 1), ... (slice <= RESCAN_SLICE entries each)
    if (jbeg >= nall) return;
    const int nq = nall - jbeg < slice ? nall - jbeg : slice;
    const int nblocks = (nq + 31) >> 5;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) lhit_n = 0;
    {
        const int* bin = bins + (size_t)c * bin_cap + jbeg;
        for (int t = threadIdx.x; t < nblocks * 32; t += 256) lbin[t] = t < nq ? bin[t] : 0;
    }
    intx4 af[KS];
    {
        const uint4* asrc = b8 + ((size_t)c * 4 + wave) * TILE_U4 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 v = asrc[s * 64];
            af[s] = *reinterpret_cast<const intx4*>(&v);
        }
    }
    const float bstep = ib.bstep[c], berr = ib.berr[c];
    const long long base = (long long)c * CHUNK_ROWS;
    const int rr0 = wave * 32 + 4 * (lane >> 5);   // + (e & 3) + 8 (e >> 2): the chunk row of accumulator element e
    float bn16[16];                                // Euclidean mode: |b~| of the lane's sixteen rows
#pragma unroll
    for (int e = 0; e < 16; ++e) bn16[e] = l2.qn ? l2.bn[base + rr0 + (e & 3) + 8 * (e >> 2)] : 0.0f;
    __syncthreads();   // (drains the loads above: from here on the vector-memory queue holds DMA only)
    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS unsigned char*)rescan_smem;
    // terms of a slot, rows of 64 dwords: [0] cand_cnt | qerr, [1] qstep | qmax, [2] qn | -   (lanes 0-31 | 32-63)
    auto issue = [&](int blk) {
        const int qi = lbin[blk * 32 + (lane & 31)];
        const unsigned slot = lds_base + (unsigned)(blk % D) * SLOT;
        if (qrows) {   // row-major int8 scan (Prepared::rows8): unit (k-step s, half h) of query qi = bytes 32 s + 16 h .. of its row
            const unsigned char* rsrc = qrows + (size_t)qi * (KS * 32) + (size_t)(2 * wave + (lane >> 5)) * 16;
#pragma unroll
            for (int t = 0; t < NG; ++t)
                glds16(rsrc + (size_t)t * 128, __builtin_amdgcn_readfirstlane(slot + (unsigned)(wave + 4 * t) * 1024u));
        } else {
        const uint4* src = q8 + (size_t)(qi >> 5) * TILE_U4 + (size_t)(2 * wave + (lane >> 5)) * 32 + (qi & 31);
#pragma unroll
        for (int t = 0; t < NG; ++t)
            glds16(src + (size_t)t * 256, __builtin_amdgcn_readfirstlane(slot + (unsigned)(wave + 4 * t) * 1024u));
        }
        const unsigned terms = slot + (unsigned)KS * 1024u;
        if (wave == 0) glds4(lane < 32 ? (const void*)(cand_cnt + qi) : (const void*)(ib.qerr + qi), __builtin_amdgcn_readfirstlane(terms));
        if (wave == 1)
            glds4(lane < 32 ? (const void*)(ib.qstep + (qi >> 7)) : (const void*)(qmax + qi), __builtin_amdgcn_readfirstlane(terms + 256u));
        if (wave == 2 && l2.qn) glds4((const void*)(l2.qn + qi), __builtin_amdgcn_readfirstlane(terms + 512u));
    };
#pragma unroll
    for (int blk = 0; blk < D - 1; ++blk)
        if (blk < nblocks) issue(blk);
    for (int blk = 0; blk < nblocks; ++blk) {          // the bin, one MFMA column block at a time
        // this wave's pieces of block blk have landed: at most the D - 2 blocks issued behind it may still be in flight
        if (D > 2 && blk + D - 2 < nblocks) wait_vmcnt<(D > 2 ? (D - 2) * NG : 0)>();
        else wait_vmcnt<0>();
        __syncthreads();                               // ... and every wave's; block blk - 1 has been read by every wave
        if (blk + D - 1 < nblocks) issue(blk + D - 1);
        const unsigned char* slot = rescan_smem + (size_t)(blk % D) * SLOT;
        const uint4* l_qf = reinterpret_cast<const uint4*>(slot);
        const int* lt = reinterpret_cast<const int*>(slot + (size_t)KS * 1024);
        const int j = blk * 32 + (lane & 31);
        const int qi = lbin[j];
        const bool live = j < nq && lt[lane & 31] >= 0;   // (-2: below the gate, -1: already with the all-pairs kernel)
        const float eq = __int_as_float(lt[32 + (lane & 31)]);
        const float sc = __int_as_float(lt[64 + (lane & 31)]) * bstep;   // the same expressions as match_rescan_kernel: the same rows pass
        const float bound = (eq * 1.0001220703125f + 1.0e-6f) + (1.0001220703125f + eq) * berr;
        const float qlow = use_gate ? gate : key_float((unsigned)lt[96 + (lane & 31)]);   // half-width pass: the hit test is the gate itself
        const float qn = l2.qn ? __int_as_float(lt[128 + (lane & 31)]) : 0.0f;
        intx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 v = l_qf[s * 64 + lane];
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s], *reinterpret_cast<const intx4*>(&v), acc, 0, 0, 0);
        }
        if (pilot_qmax) {
            // VFM_RECORDS_MX6_PILOT, in front of the selection: the bin holds the queries whose best fp6 chunk this is; the chunk's best
            // EXACT integer score gives a lower bound of the query's exact maximum that carries the int8 image's bound (~0.01) instead of
            // the fp6 image's (~0.06): it raises qmax, which the selection and the rescans behind it test against.  No hits are listed.
            int smax = -0x7fffffff;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = rr0 + (e & 3) + 8 * (e >> 2);
                if (base + rr < m) smax = max(smax, acc[e]);
            }
            smax = max(smax, __shfl_xor(smax, 32));   // the two half-waves hold the same query's other rows
            if (j < nq && lane < 32 && smax > -0x7fffffff) atomicMax(&pilot_qmax[qi], float_key((sc * (float)smax - bound) - 2.0e-6f));
            __syncthreads();
            continue;
        }
        unsigned hits = 0u;
        if (live) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = rr0 + (e & 3) + 8 * (e >> 2);
                float up = sc * (float)acc[e] + bound;
                if (l2.qn) up = l2_upper_row(qn, bn16[e], up, l2.slack);
                if (base + rr < m && up >= qlow) hits |= 1u << e;
            }
        }
        if (hits) {
            int at = atomicAdd(&lhit_n, __popc(hits));
            while (hits) {
                const int e = __ffs(hits) - 1;
                hits &= hits - 1u;
                const int rr = rr0 + (e & 3) + 8 * (e >> 2);
                // (a Euclidean search's bound is not symmetric about the score: its rows are never filtered -- +Inf)
                const float upe = l2.qn ? __builtin_inff() : sc * (float)acc[e] + bound;
                if (at < RESCAN_LHITS) {
                    lhq[at] = qi;
                    lhr[at] = (unsigned char)rr;
                    lhu[at] = upe;
                } else {   // more hits than the staging buffer holds (duplicate-rich chunk): on the spot
                    const int pos = cand_cnt[qi] + (int)atomicAdd(&hit_cnt[(size_t)qi * BIN_CNT_STRIDE], 1u);
                    if (pos < cap) {
                        cand[(size_t)qi * cap + pos] = ((unsigned)c << 8) | (unsigned)rr;
                        cand_up[(size_t)qi * cap + pos] = upe;
                    }
                }
                ++at;
            }
        }
        // (a full staging buffer is emptied before the next block adds to it; lhit_n is read by every thread between two barriers
        // that no atomic of another block can cross, so the branch is uniform)
        __syncthreads();
        if (lhit_n > RESCAN_LHITS / 2 || blk + 1 == nblocks) {
            wait_vmcnt<0>();   // (the atomics below are compiler-tracked: nothing of the ring may be pending behind them)
            const int nh = lhit_n < RESCAN_LHITS ? lhit_n : RESCAN_LHITS;
            for (int i = threadIdx.x; i < nh; i += 256) {
                // (the list's length stays as the selection / match_rescan_kernel left it during this kernel; the appended rows are
                // counted in hit_cnt, a line of their own per query, and added by match_rescan_close_kernel)
                const int hq = lhq[i];
                const int pos = cand_cnt[hq] + (int)atomicAdd(&hit_cnt[(size_t)hq * BIN_CNT_STRIDE], 1u);
                if (pos < cap) {
                    cand[(size_t)hq * cap + pos] = ((unsigned)c << 8) | (unsigned)lhr[i];
                    cand_up[(size_t)hq * cap + pos] = lhu[i];
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) lhit_n = 0;
        }
    }
}

template <int KS>
int launch_rescan_chunk_ks(const SearchWs& w, int nchunks, int64_t n, int64_t m, I8Bounds ib, const Prepared& Q, const Prepared& B,
                           int use_gate, float gate, const int* guard, L2Terms l2, hipStream_t st, bool pilot) {
    const size_t lds = (size_t)rescan_ring_depth<KS>() * rescan_slot_bytes<KS>();   // + ~14 KB of static LDS: past 64 KB in all from KS = 16
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_rescan_chunk_kernel<KS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_mark(attr_set);
    }
    // Short workgroups: a workgroup that walks a whole bin (~600 queries on lifted descriptors) lives ~80 us on 160 registers x 4 waves
    // and 61 KB, and while a grid of those is resident nothing as register-heavy as a ViT GEMM wave (156 - 416 registers) is placed
    // beside it -- in C3 as a pipeline the feature stage of the next pair stood still behind this kernel (tools/trace_c3_pipe.sh).
    const int slice = vfm_cfg().finish_short ? 128 : RESCAN_SLICE;
    hipLaunchKernelGGL(match_rescan_chunk_kernel<KS>, dim3((unsigned)nchunks, (unsigned)((w.bin_cap + slice - 1) / slice)),
                       dim3(256), lds, st, n, m, ib, (const uint4*)Q.tiles8, (const uint4*)B.tiles8, (const unsigned*)w.qmax, w.cand_cnt,
                       w.cand, w.cap, (const unsigned*)w.bin_cnt, (const int*)w.bins, use_gate, gate, guard, l2, w.bin_cap, w.hit_cnt, w.cand_up,
                       slice, vfm_cfg().rescan_rows ? (const unsigned char*)Q.rows8 : (const unsigned char*)nullptr, pilot ? w.qmax : (unsigned*)nullptr);
    VFM_CHECK_LAUNCH("match_rescan_chunk_kernel");
    return VFM_OK;
}
int launch_rescan_chunk(const SearchWs& w, int nchunks, int64_t n, int64_t m, int d, I8Bounds ib, const Prepared& Q, const Prepared& B,
                        int use_gate, float gate, const int* guard, L2Terms l2, hipStream_t st, bool pilot = false) {
    switch (d / 32) {
        case 8: return launch_rescan_chunk_ks<8>(w, nchunks, n, m, ib, Q, B, use_gate, gate, guard, l2, st, pilot);
        case 12: return launch_rescan_chunk_ks<12>(w, nchunks, n, m, ib, Q, B, use_gate, gate, guard, l2, st, pilot);
        case 16: return launch_rescan_chunk_ks<16>(w, nchunks, n, m, ib, Q, B, use_gate, gate, guard, l2, st, pilot);
        case 20: return launch_rescan_chunk_ks<20>(w, nchunks, n, m, ib, Q, B, use_gate, gate, guard, l2, st, pilot);
        default: return launch_rescan_chunk_ks<24>(w, nchunks, n, m, ib, Q, B, use_gate, gate, guard, l2, st, pilot);
    }
}

// VFM_RECORDS_MX6_PILOT: every query (zero rows aside) into the bin of the chunk its best fp6 score came from
__global__ __launch_bounds__(256) void match_pilot_bin_kernel(int64_t n, const unsigned long long* __restrict__ qbest,
                                                              const float* __restrict__ invq, unsigned* __restrict__ bin_cnt,
                                                              int* __restrict__ bins, int bin_cap, int nchunks) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n || invq[q] == 0.0f) return;
    const unsigned long long v = qbest[q];
    if ((v >> 32) == 0ull) return;   // no un-padded chunk gave a lower bound
    const unsigned c = (unsigned)v;
    if (c >= (unsigned)nchunks) return;
    const unsigned pos = atomicAdd(&bin_cnt[(size_t)c * BIN_CNT_STRIDE], 1u);
    if (pos < (unsigned)bin_cap) bins[(size_t)c * bin_cap + pos] = (int)q;   // (a bin that is full: those queries keep the fp6 lower bound)
}

// ---------------------------------------------------------------------------------------------
// Half-width pass: the device-side guard (VERDICT r2 / ADVICE r2: "no 171 ms registration, ever").
// half_guard_kernel (one workgroup, after the selection): the search's survivor count -- fb_count[5] from match_select_half_kernel,
// or the sum of the bin counts in the fused form, which it stores there -- against HALF_GUARD_PER_QUERY * n; above it (or where the
// fused coarse kernel saturated a bin) fb_count[HALF_GUARD_FLAG] = 1 and the selection's all-pairs fallbacks are withdrawn.
// Then match_rescan_kernel empties every list, match_rescan_chunk_kernel returns at once, and match_gatepass_kernel resolves
// the search: one workgroup per map chunk, wave w holds tile w of the chunk as the MFMA's first operand (48 registers at
// d = 384), the scan's int8 tiles stream past it (next tile's fragments loaded under the current tile's MFMAs), and every row
// whose full-width upper bound  s_q s_c S + A + B_c  reaches the gate -- the hit test of the rescans -- is appended to its
// query's list.  The matrix work of the full-width pass in a kernel light enough to sit beside anything (no LDS, <= 2 waves
// per SIMD), launched behind every half-width selection and returning at once unless the flag is up.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void half_guard_kernel(int* __restrict__ fb_count, const unsigned* __restrict__ bin_cnt, int nchunks,
                                                         int fused, long long limit) {
    __shared__ long long part[4];
    long long s = 0;
    if (fused)
        for (int c = threadIdx.x; c < nchunks; c += 256) s += (long long)bin_cnt[(size_t)c * BIN_CNT_STRIDE];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane_id() == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long total = fused ? part[0] + part[1] + part[2] + part[3] : (long long)fb_count[5];
        // (the fused fp6 kernel raises the flag itself when a workgroup's list overflows, and its survivors are then not binned: the
        // load figure must still say "far too many")
        if (fused && fb_count[HALF_GUARD_FLAG] != 0 && total < 0x3FFFFFFFll) total = 0x3FFFFFFFll;
        if (fused) fb_count[5] = (int)(total > 0x7FFFFFFFll ? 0x7FFFFFFFll : total);   // the search's load figure
        if (total > limit || fb_count[HALF_GUARD_FLAG] != 0) {
            fb_count[HALF_GUARD_FLAG] = 1;
            fb_count[0] = 0;   // queries the selection sent to the all-pairs kernel: the gate pass decides them too
        }
    }
}

template <int KS>  // k-steps of 32 columns (d / 32)
__global__ __launch_bounds__(256, 2) void match_gatepass_kernel(int64_t n, int64_t m, int nq_tiles, I8Bounds ib,
                                                                const float* __restrict__ invq, const uint4* __restrict__ q8,
                                                                const uint4* __restrict__ b8, float gate, int* __restrict__ cand_cnt,
                                                                unsigned* __restrict__ cand, int cap, const int* __restrict__ guard,
                                                                unsigned* __restrict__ qlow) {
    // qlow (the search's qmax array: unused by the half-width selection, zero = -Inf): float_key of a lower bound of the query's
    // exact maximum, published by whichever workgroup finds a better one.  Where hundreds of rows per query reach the gate -- lifted
    // descriptors with a common component -- a row whose upper bound lies below it cannot be the arg-max and is not appended (the
    // arg-max itself, and every row tied with it, has an upper bound at or above every lower bound ever published): the lists the
    // refinement gets shrink from "every row above the gate" to "rows near the best" (forced half-width pass on such a map: 22 ms
    // per C2-size registration before)
    if (*guard == 0) return;
    constexpr int TILE_U4 = KS * 64;
    const int c = blockIdx.x, lane = lane_id(), wave = threadIdx.x >> 6;
    intx4 af[KS];
    {
        const uint4* asrc = b8 + ((size_t)c * 4 + wave) * TILE_U4 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 v = asrc[s * 64];
            af[s] = *reinterpret_cast<const intx4*>(&v);
        }
    }
    const float sb = ib.bstep[c], be = ib.berr[c];
    const long long row0 = (long long)c * CHUNK_ROWS + wave * 32 + 4 * (lane >> 5);   // + (e & 3) + 8 (e >> 2): accumulator element e
    constexpr bool AHEAD = KS <= 12;   // 3 x 4 KS fragment registers fit two waves per SIMD up to d = 384
    uint4 qn[AHEAD ? KS : 1];
    if constexpr (AHEAD) {
        const uint4* qsrc = q8 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) qn[s] = qsrc[s * 64];
    }
    for (int qt = 0; qt < nq_tiles; ++qt) {
        intx4 qf[KS];
        if constexpr (AHEAD) {
#pragma unroll
            for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const intx4*>(&qn[s]);
            if (qt + 1 < nq_tiles) {   // the next tile's fragments, under this tile's MFMAs
                const uint4* qsrc = q8 + (size_t)(qt + 1) * TILE_U4 + lane;
#pragma unroll
                for (int s = 0; s < KS; ++s) qn[s] = qsrc[s * 64];
            }
        } else {
            const uint4* qsrc = q8 + (size_t)qt * TILE_U4 + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const uint4 v = qsrc[s * 64];
                qf[s] = *reinterpret_cast<const intx4*>(&v);
            }
        }
        const int64_t q = (int64_t)qt * 32 + (lane & 31);
        const float eq = ib.qerr[q], sq = ib.qstep[q >> 7];
        const bool live = q < n && invq[q] != 0.0f;
        intx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[s], qf[s], acc, 0, 0, 0);
        int mx = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = max(mx, acc[r]);
        // the expressions of match_rescan_chunk_kernel / match_rescan_kernel: the same rows pass
        const float sc = sq * sb, bound = (eq * 1.0001220703125f + 1.0e-6f) + (1.0001220703125f + eq) * be;
        float thr = gate;
        if (live) {
            const float ql = key_float(__hip_atomic_load(qlow + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            // this lane's best row lies at or above sc * mx - bound (padding rows of the last chunk score 0: only a positive score counts)
            const float mine = (row0 + 27 < m || mx > 0) ? __builtin_fmaf(sc, (float)mx, -bound) : -__builtin_inff();
            if (mine > ql && mine >= gate) atomicMax(qlow + q, float_key(mine));
            thr = fmaxf(gate, fmaxf(ql, mine));
        }
        const bool any = live && (sc * (float)mx + bound >= thr);
        if (__ballot(any) != 0ull && any) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long row = row0 + (e & 3) + 8 * (e >> 2);
                if (row < m && sc * (float)acc[e] + bound >= thr) {
                    const int pos = atomicAdd(&cand_cnt[q], 1);
                    if (pos < cap) cand[(size_t)q * cap + pos] = ((unsigned)c << 8) | (unsigned)(row - (long long)c * CHUNK_ROWS);
                }
            }
        }
    }
}

// after the rescans: lists longer than their capacity go to the all-pairs kernel, crowded ones to the fp32 refinement's work
// list -- appended with ONE atomic per wavefront (on duplicate-rich maps every query is crowded: 20 000 same-address atomics
// took 190 us)
__global__ __launch_bounds__(256) void match_rescan_close_kernel(int64_t n, int* __restrict__ cand_cnt, int cap,
                                                                 int* __restrict__ fb_count, int* __restrict__ fb_list,
                                                                 int* __restrict__ todo, const float* __restrict__ invq_half,
                                                                 const unsigned* __restrict__ hit_cnt) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = lane_id();
    int cnt = q < n ? cand_cnt[q] : 0;
    if (q < n && cnt >= 0) {   // + the rows of the chunk-major rescan
        const int hits = (int)hit_cnt[(size_t)q * BIN_CNT_STRIDE];
        if (hits) cand_cnt[q] = cnt = cnt + hits;
    }
    // half-width pass (invq_half != NULL): a live query whose surviving chunks held no row at the gate has no match
    if (invq_half && q < n && cnt == 0 && invq_half[q] != 0.0f) cand_cnt[q] = -2;
    if (cnt > cap) {
        cand_cnt[q] = -1;
        fb_list[atomicAdd(fb_count, 1)] = (int)q;
    }
    const bool crowded = cnt >= REFINE_MIN_I8 && cnt <= cap;
    const unsigned long long bal = __ballot(crowded);
    if (bal == 0ull) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(fb_count + 6, __popcll(bal));
    base = __shfl(base, 0);
    if (crowded) todo[base + __popcll(bal & ((1ull << lane) - 1ull))] = (int)q;
}

// ---------------------------------------------------------------------------------------------
// fp32 refinement of crowded candidate lists (near-duplicate map rows).
//
// Real lifted descriptors are bilinear interpolations of a 16 x 21 patch grid (image_features.py:104-110,
// prepare_scenes.py:85-104): neighbouring map points differ by less than the fp16 window (2.5e-3), so a query
// can have dozens of candidate chunks, and chunks whose two best rows are both inside the window.  Deciding
// all of them in fp64 (or, past the old 40-entry cap, all M rows) was a cliff.  Here one wavefront per such
// query scores every candidate row in fp32 -- 16 lanes per row, 4 rows per pass, each lane a sequential fma
// chain over d/16 elements followed by a 4-level xor tree -- and keeps only the rows within
//     w2 = 2 * (d/16 + 4 + 2) * 2^-24      (3.6e-6 at d = 384)
// of the fp32 maximum.  Proof that the oracle's arg-max survives: the fp32 value s of a row differs from the
// exact dot product t of the same fp32-normalised rows (the oracle's definition) by at most
// gamma = (d/16 + 4) u * sum|q_k b_k| <= (d/16 + 4) u (1 + 1e-6), u = 2^-24 (one rounding per fma / add along the
// longest path of the summation tree; Cauchy-Schwarz on unit rows).  With j* the exact arg-max and j' the fp32
// arg-max: s(j*) >= t(j*) - gamma >= t(j') - gamma >= s(j') - 2 gamma; every row that ties with j* exactly is
// inside the same margin, so the fp64 decision (ties -> lowest index) sees them all.  The surviving rows
// replace the query's list as single-row entries: match_rescore_kernel is unchanged.
// ---------------------------------------------------------------------------------------------
// state of one wavefront refining one query: the fp32-normalised query in registers, the kept rows in LDS
// (NT = float4 per lane the arrays hold, >= d / 64: 6 at d = 384 -- 24 + 2 x 25 registers instead of 48 + 2 x 49, which is what lets the
// refinement run four waves per SIMD instead of two)
template <int NT>
struct RefineWave {
    Rows b;
    const float* invb;
    int d, nt, g, l, lane;
    float w2;
    float4 qv[NT];
    unsigned* lrow;
    float* lsc;
    int kept;        // wave-uniform
    float runmax;    // wave-uniform
    bool overflow;

    __device__ __forceinline__ void init(Rows q, float iq, int64_t qi, Rows b_, const float* invb_, int d_, float w2_,
                                         unsigned* lrow_, float* lsc_) {
        b = b_; invb = invb_; d = d_; w2 = w2_; lrow = lrow_; lsc = lsc_;
        lane = lane_id();
        g = lane >> 4;   // row slot of the pass
        l = lane & 15;   // k-slice
        nt = d >> 6;     // float4 per lane (d % 64 == 0)
        kept = 0;
        runmax = -3.0e38f;
        overflow = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t < nt) {
                float4 v = q.ld4(qi * (int64_t)d + 4 * (l + 16 * t));
                v.x = v.x * iq; v.y = v.y * iq; v.z = v.z * iq; v.w = v.w * iq;  // the fp32-normalised query (faiss' xq)
                qv[t] = v;
            }
        }
    }
    // fp32 score of this lane group's row (row < 0: none); identical in the 16 lanes of the group
    __device__ __forceinline__ float score4(long long row) const {
        float acc = 0.0f;
        if (row >= 0) {
            const float ib = invb[row];
            const int64_t br = row * (int64_t)d + 4 * l;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < nt) {
                    const float4 bv = b.ld4(br + 64 * t);
                    acc = __builtin_fmaf(qv[t].x, bv.x * ib, acc);
                    acc = __builtin_fmaf(qv[t].y, bv.y * ib, acc);
                    acc = __builtin_fmaf(qv[t].z, bv.z * ib, acc);
                    acc = __builtin_fmaf(qv[t].w, bv.w * ib, acc);
                }
            }
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off);
        return acc;
    }
    // score4 in two halves, so that the next pass's row can be on its way while this pass's is summed (the same operations in
    // the same order: the same float)
    struct Row {
        float4 v[NT];
        float ib;
    };
    __device__ __forceinline__ void load4(Row& r, long long row) const {
        if (row >= 0) {
            r.ib = invb[row];
            const int64_t br = row * (int64_t)d + 4 * l;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t < nt) r.v[t] = b.ld4(br + 64 * t);
        }
    }
    __device__ __forceinline__ float dot4(const Row& r, long long row) const {
        float acc = 0.0f;
        if (row >= 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < nt) {
                    acc = __builtin_fmaf(qv[t].x, r.v[t].x * r.ib, acc);
                    acc = __builtin_fmaf(qv[t].y, r.v[t].y * r.ib, acc);
                    acc = __builtin_fmaf(qv[t].z, r.v[t].z * r.ib, acc);
                    acc = __builtin_fmaf(qv[t].w, r.v[t].w * r.ib, acc);
                }
            }
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off);
        return acc;
    }
    // one pass: (row, sc) of the four lane groups; keeps the rows within w2 of the running fp32 maximum
    __device__ __forceinline__ void consider(long long row, float sc) {
        float pm = (row >= 0) ? sc : -3.0e38f;
        pm = fmaxf(pm, __shfl_xor(pm, 16));
        pm = fmaxf(pm, __shfl_xor(pm, 32));
        if (pm > runmax) {  // prune the kept list against the new maximum
            runmax = pm;
            const float thr = runmax - w2;
            const bool mine = lane < kept && lsc[lane] >= thr;
            const unsigned r = lane < kept ? lrow[lane] : 0u;
            const float sv = lane < kept ? lsc[lane] : 0.f;
            const unsigned long long bal = __ballot(mine);
            __builtin_amdgcn_wave_barrier();
            if (mine) {
                const int pos = __popcll(bal & ((1ull << lane) - 1ull));
                lrow[pos] = r;
                lsc[pos] = sv;
            }
            __builtin_amdgcn_wave_barrier();
            kept = __popcll(bal);
        }
        const float thr = runmax - w2;
        const bool add = (l == 0) && row >= 0 && sc >= thr;
        const unsigned long long bal = __ballot(add);
        if (add) {
            const int pos = kept + __popcll(bal & ((1ull << lane) - 1ull));
            if (pos < REFINE_KEEP) {
                lrow[pos] = (unsigned)row;
                lsc[pos] = sc;
            }
        }
        kept += __popcll(bal);
        if (kept > REFINE_KEEP) {
            overflow = true;
            kept = REFINE_KEEP;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // result: the kept rows become the query's single-row candidate list (or the query goes to the all-pairs kernel)
    __device__ __forceinline__ void finish(int64_t qi, unsigned* mycand, int* cand_cnt, int* fb_count, int* fb_list, int stats) {
        __builtin_amdgcn_wave_barrier();
        if (overflow) {  // > REFINE_KEEP rows tie within the fp32 margin: the all-pairs kernel decides
            if (lane == 0) {
                cand_cnt[qi] = -1;
                const int slot = atomicAdd(fb_count, 1);
                fb_list[slot] = (int)qi;
            }
            return;
        }
        if (stats && lane == 0) {  // statistics: [1] queries refined, [3] rows they keep
            atomicAdd(fb_count + 1, 1);
            atomicAdd(fb_count + 3, kept);
        }
        if (lane < kept) {
            const unsigned row = lrow[lane];
            mycand[lane] = ((row / CHUNK_ROWS) << 8) | (row % CHUNK_ROWS);
        }
        if (lane == 0) cand_cnt[qi] = kept;
    }
};

// int8 pass: which rows of a query's candidate chunks matter.  One wave per query; per candidate chunk the exact integer
// scores of its 128 rows against the query (v_dot4 over the chunk's four int8 tiles: 48 KB, contiguous, L2 / Infinity-Cache
// resident -- the whole int8 map is 77 MB at C2 -- instead of 196 KB of fp32 rows); every row whose upper bound reaches the
// query's lower bound replaces the chunk entries as a single-row entry.  Few registers on purpose (the loop is latency-bound:
// 8 x 16 bytes per lane in flight, 8 waves per SIMD); match_refine_kernel / match_rescore_kernel then see the lists the fp16
// pass would have produced.
__global__ __launch_bounds__(256) void match_rescan_kernel(int64_t n, int64_t m, int d, I8Bounds ib, const uint4* __restrict__ q8,
                                                           const uint4* __restrict__ b8, const unsigned* __restrict__ qmax,
                                                           int* __restrict__ cand_cnt, unsigned* __restrict__ cand, int cap,
                                                           unsigned* __restrict__ hits, int hcap, int* __restrict__ fb_count,
                                                           int* __restrict__ fb_list, int use_gate, float gate,
                                                           const int* __restrict__ guard, L2Terms l2) {
    __shared__ uint4 l_q8[4][48];  // the query's int8 row, unit by unit (d <= 768)
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
    if (qi >= n) return;
    if (guard && *guard) {   // half-width pass, too many survivors: the lists start empty for match_gatepass_kernel's row hits
        if (lane == 0) cand_cnt[qi] = 0;
        return;
    }
    const int cnt = cand_cnt[qi];
    if (cnt <= 0) return;  // zero query / below the gate (-2) / overflow (-1)
    if (cnt > cap) {       // (fused half-width pass: more bin overflows than the list holds) the all-pairs kernel decides
        if (lane == 0) {
            cand_cnt[qi] = -1;
            fb_list[atomicAdd(fb_count, 1)] = (int)qi;
        }
        return;
    }
    unsigned* mycand = cand + (size_t)qi * cap;
    unsigned* myhits = hits + (size_t)qi * hcap;
    const int units8 = d >> 4;  // 16-byte units per int8 row
    if (lane < units8) l_q8[wave][lane] = q8[(size_t)(qi >> 5) * (units8 * 32) + (size_t)lane * 32 + (qi & 31)];
    const float eq = ib.qerr[qi];
    const float sq = ib.qstep[qi >> 7], A = eq * 1.0001220703125f + 1.0e-6f, mult = 1.0001220703125f + eq;
    const float qlow = use_gate ? gate : key_float(qmax[qi]);   // half-width pass: the hit test is the gate itself
    const float qn_l2 = l2.qn ? l2.qn[qi] : 0.0f;
    __builtin_amdgcn_wave_barrier();
    int nhit = 0;  // wave-uniform
    // Up to 64 entries (nearly every query): they sit in registers, one per lane, before the first hit is written, so the
    // hits go straight into the query's list.  Longer lists collect their hits in the scratch list and copy them back.
    const bool direct = cnt <= 64;
    unsigned* out = direct ? mycand : myhits;
    const int ocap = direct ? cap : hcap;
    for (int e0 = 0; e0 < cnt; e0 += 64) {
        const unsigned batch = (e0 + lane < cnt) ? mycand[e0 + lane] : 0u;
        const int nb = cnt - e0 < 64 ? cnt - e0 : 64;
        for (int j = 0; j < nb; ++j) {
            const unsigned entry = __shfl(batch, j);  // wave-uniform
            const int c = (int)(entry >> 8);
            if (!(entry & 128u)) {  // a single-row entry (top-2 records: the chunk's second-best cannot reach the bound)
                if (lane == 0 && nhit < ocap) out[nhit] = entry;
                ++nhit;
                continue;
            }
            const long long base = (long long)c * CHUNK_ROWS;
            const float sc = sq * ib.bstep[c], bound = A + mult * ib.berr[c];
            // rows `lane` and `lane + 64` of the chunk (tile rr >> 5, position rr & 31); the loads of both go out together:
            // 16 x 16 bytes per lane in flight, three round trips per chunk at d = 384 (the loop is latency-bound)
            const uint4* src0 = b8 + ((size_t)c * 4 + (lane >> 5)) * (size_t)(units8 * 32) + (lane & 31);
            const uint4* src1 = src0 + 2 * (size_t)(units8 * 32);
            int acc2[2] = {0, 0};
            for (int u0 = 0; u0 < units8; u0 += 8) {  // units8 = 16, 24, 32, 40 or 48
                uint4 bv0[8], bv1[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    bv0[k] = src0[(u0 + k) * 32];
                    bv1[k] = src1[(u0 + k) * 32];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint4 qv = l_q8[wave][u0 + k];
                    acc2[0] = __builtin_amdgcn_sdot4((int)bv0[k].x, (int)qv.x, acc2[0], false);
                    acc2[0] = __builtin_amdgcn_sdot4((int)bv0[k].y, (int)qv.y, acc2[0], false);
                    acc2[0] = __builtin_amdgcn_sdot4((int)bv0[k].z, (int)qv.z, acc2[0], false);
                    acc2[0] = __builtin_amdgcn_sdot4((int)bv0[k].w, (int)qv.w, acc2[0], false);
                    acc2[1] = __builtin_amdgcn_sdot4((int)bv1[k].x, (int)qv.x, acc2[1], false);
                    acc2[1] = __builtin_amdgcn_sdot4((int)bv1[k].y, (int)qv.y, acc2[1], false);
                    acc2[1] = __builtin_amdgcn_sdot4((int)bv1[k].z, (int)qv.z, acc2[1], false);
                    acc2[1] = __builtin_amdgcn_sdot4((int)bv1[k].w, (int)qv.w, acc2[1], false);
                }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int rr = lane + 64 * half;
                const int acc = acc2[half];
                float up = sc * (float)acc + bound;
                if (l2.qn) up = l2_upper_row(qn_l2, l2.bn[base + rr], up, l2.slack);
                const bool hit = base + rr < m && up >= qlow;
                const unsigned long long bal = __ballot(hit);
                if (hit) {
                    const int pos = nhit + __popcll(bal & ((1ull << lane) - 1ull));
                    if (pos < ocap) out[pos] = ((unsigned)c << 8) | (unsigned)rr;
                }
                nhit += __popcll(bal);
            }
        }
    }
    if (nhit > ocap || nhit > cap) {  // more rows inside the bounds than a list holds: the all-pairs kernel decides
        if (lane == 0) {
            cand_cnt[qi] = -1;
            const int slot = atomicAdd(fb_count, 1);
            fb_list[slot] = (int)qi;
        }
        return;
    }
    if (!direct) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");  // the hits written above are read back by other lanes
        for (int i = lane; i < nhit; i += 64) mycand[i] = myhits[i];
    }
    if (lane == 0) {
        cand_cnt[qi] = nhit;
        // (match_rescan_close_kernel builds the refinement's work list from the final counts: lists of eight rows or more;
        // up to seven rows go straight to the fp64 decision -- a handful of fp64 dot products costs less than the latency
        // of one refinement wave)
    }
}

// dense records: the candidate entries of match_select_kernel (int8 pass: as rewritten by match_rescan_kernel)
template <int NT>
__global__ __launch_bounds__(256) void match_refine_kernel(Rows q, const float* __restrict__ invq,
                                                           Rows b, const float* __restrict__ invb,
                                                           int64_t n, int64_t m, int d, float w2, int* __restrict__ cand_cnt,
                                                           unsigned* __restrict__ cand, int cap, int* __restrict__ fb_count,
                                                           int* __restrict__ fb_list, int stats, const int* __restrict__ todo,
                                                           const int* __restrict__ todo_count, const float* __restrict__ cand_up,
                                                           const unsigned* __restrict__ hit_cnt, I8Bounds ib) {
    __shared__ unsigned l_row[4][REFINE_KEEP];
    __shared__ float l_sc[4][REFINE_KEEP];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    // todo != NULL (int8 pass): the queries match_rescan_close_kernel found crowded, a list whose length is on the device -- one wave
    // per possible entry, waves past its end leave at once (the loop serves a caller that launches fewer); otherwise one wave per query
    const int64_t slot0 = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nslots = todo ? (int64_t)*todo_count : n;
  for (int64_t slot = slot0; slot < nslots; slot += (int64_t)gridDim.x * 4) {
    const int64_t qi = todo ? (int64_t)todo[slot] : slot;
    if (qi >= n) return;
    // Everything that depends on the query's index alone is requested at once -- the list's length, how much of it the chunk-major
    // rescan appended, the query's E and 1 / |q|, the first 64 entries with their upper bounds -- and the query's row (R.init) right
    // behind: a query had been a chain of seven dependent round trips (length -> entries -> flags -> appended count -> bounds ->
    // entries again -> rows), 60 us per query at 9891 queries on 4096 waves.
    unsigned* mycand = cand + (size_t)qi * cap;
    const float* myup = cand_up ? cand_up + (size_t)qi * cap : nullptr;
    const int cnt = cand_cnt[qi];
    const int hits = cand_up ? (int)hit_cnt[(size_t)qi * BIN_CNT_STRIDE] : 0;
    const float eq = cand_up ? ib.qerr[qi] : 0.0f;
    const float iq = invq[qi];
    const unsigned ce0 = mycand[lane];                  // (cap >= 64: the first 64 slots exist whatever the length)
    const float up0 = cand_up ? myup[lane] : 0.0f;
    if (cnt <= 0) continue;  // zero query / nothing / overflow (-1: the all-pairs kernel decides)
    // wave-uniform: is this list crowded?
    bool flagged = lane < cnt && (ce0 & 128u) != 0u;
    for (int e = 64 + lane; e < cnt; e += 64) flagged |= (mycand[e] & 128u) != 0u;
    if (cnt < REFINE_MIN && !__any(flagged)) continue;
    RefineWave<NT> R;
    R.init(q, iq, qi, b, invb, d, w2, l_row[wave], l_sc[wave]);
    // The rows the chunk-major rescan appended (the last hit_cnt entries of the list) carry the upper bound U = s_q s_c S + A + B_c of
    // their exact score (cand_up), S their exact integer score; U - 2 (A + B_c) is a LOWER bound of the same score.  The rescans tested
    // a row against the coarse pass's lower bound of the query's maximum -- the fp6 image's, 0.06 below the maximum on unit rows, so
    // that 45 rows per query came through on lifted descriptors where 13 lie inside the int8 bounds --; here every
    // row is tested against the best lower bound among the list's own rows, and the rest never has its 1.5 KB fp32 row read.  A row
    // dropped has U < L <= the exact score of another row of the list: it is not the arg-max and does not tie with it.
    int split = cnt;        // first appended entry
    float lowest = -__builtin_inff();
    if (cand_up) {
        split = cnt - hits < 0 ? 0 : cnt - hits;
        const float A = eq * 1.0001220703125f + 1.0e-6f, mult = 1.0001220703125f + eq;
        if (lane >= split && lane < cnt) {
            const float bnd = A + mult * ib.berr[ce0 >> 8];
            if (up0 < 3.0e38f) lowest = (up0 - 2.0f * bnd) - 4.0e-6f;   // (fp32 evaluation of U and of this line: a few 1e-7)
        }
        for (int e = (split > 64 ? split : 64) + lane; e < cnt; e += 64) {
            const float u = myup[e];
            const float bnd = A + mult * ib.berr[mycand[e] >> 8];
            if (u < 3.0e38f) lowest = fmaxf(lowest, (u - 2.0f * bnd) - 4.0e-6f);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) lowest = fmaxf(lowest, __shfl_xor(lowest, off));
    }
    // single-row entries: 4 per pass, 64 list entries per load, the next pass's rows in flight under the current pass's sums (a
    // pass had been two dependent round trips -- the entry, then the row --: 164 us at the 25 rows per query of lifted descriptors)
    for (int e0 = 0; e0 < cnt; e0 += 64) {
        const int nblk = cnt - e0 < 64 ? cnt - e0 : 64;
        long long myrow = -1;
        if (lane < nblk) {
            const unsigned ce = e0 == 0 ? ce0 : mycand[e0 + lane];
            if (!(ce & 128u)) {
                myrow = (long long)(ce >> 8) * CHUNK_ROWS + (ce & 127u);
                if (myrow >= m) myrow = -1;
                if (e0 + lane >= split && (e0 == 0 ? up0 : myup[e0 + lane]) < lowest) myrow = -1;
            }
        }
        if (!__any(myrow >= 0)) continue;
        // the rows that are left, packed to the front (the dropped ones are scattered through the list: a pass of four costs its
        // round trip as long as one of its rows is live)
        const unsigned long long keep = __ballot(myrow >= 0);
        const int nkeep = __popcll(keep);
        {
            // lane l takes the row of the l-th live lane: the position of the l-th set bit of `keep`, by halving
            int need = lane, src = 0;
            unsigned long long rest = keep;
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) {
                const unsigned long long lowmask = (1ull << sh) - 1ull;
                const int below = __popcll(rest & lowmask);
                if (need >= below) {
                    need -= below;
                    rest >>= sh;
                    src += sh;
                } else {
                    rest &= lowmask;
                }
            }
            const long long packed = __shfl(myrow, src & 63);
            myrow = lane < nkeep ? packed : -1;
        }
        const int nblk_keep = nkeep;
        auto row_at = [&](int p) { return p < nblk_keep ? __shfl(myrow, (p + R.g) & 63) : -1ll; };   // (lanes >= nkeep hold -1)
        typename RefineWave<NT>::Row ra, rb;
        long long rowa = row_at(0), rowb;
        R.load4(ra, rowa);
        for (int p = 0; p < nblk_keep; p += 8) {
            rowb = row_at(p + 4);
            R.load4(rb, rowb);
            if (__any(rowa >= 0)) R.consider(rowa, R.dot4(ra, rowa));
            rowa = row_at(p + 8);
            R.load4(ra, rowa);
            if (__any(rowb >= 0)) R.consider(rowb, R.dot4(rb, rowb));
        }
    }
    // whole-chunk entries: all 128 rows of the chunk
    for (int e = 0; e < cnt; ++e) {
        const unsigned ce = mycand[e];  // wave-uniform
        if (!(ce & 128u)) continue;
        const long long base = (long long)(ce >> 8) * CHUNK_ROWS;
        for (int r0 = 0; r0 < CHUNK_ROWS; r0 += 4) {
            long long row = base + r0 + R.g;
            if (row >= m) row = -1;
            if (__any(row >= 0)) R.consider(row, R.score4(row));
        }
    }
    if (R.overflow) continue;  // > REFINE_KEEP rows tie within the fp32 margin (exact duplicates): the list stays as
                               // match_select_kernel wrote it and match_rescore_kernel decides all of it in fp64
    R.finish(qi, mycand, cand_cnt, fb_count, fb_list, stats);
    __builtin_amdgcn_wave_barrier();
  }
}

// sparse records (match_coarse_pipe_kernel<., true>): filter the query's records against its FINAL coarse maximum, then
// refine in fp32 if more than two rows remain.  Replaces match_select_kernel + match_refine_kernel; one wave per query.
__global__ __launch_bounds__(256) void match_filter_refine_kernel(Rows q, const float* __restrict__ invq,
                                                                  Rows b, const float* __restrict__ invb,
                                                                  int64_t n, int64_t m, int d, float window, float w2,
                                                                  const unsigned* __restrict__ qmax,
                                                                  const unsigned* __restrict__ rec_cnt, const uint2* __restrict__ rec,
                                                                  int rcap, int* __restrict__ cand_cnt, unsigned* __restrict__ cand,
                                                                  int cap, int* __restrict__ fb_count, int* __restrict__ fb_list, int stats) {
    __shared__ unsigned l_row[4][REFINE_KEEP];
    __shared__ float l_sc[4][REFINE_KEEP];
    __shared__ unsigned l_cand[4][FILTER_LDS_ROWS];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
    if (qi >= n) return;
    unsigned* mycand = cand + (size_t)qi * cap;
    const float iq = invq[qi];
    if (iq == 0.0f) {  // zero query row: decided directly by match_rescore_kernel (index 0, score 0)
        if (lane == 0) cand_cnt[qi] = 0;
        return;
    }
    const unsigned total = rec_cnt[qi];
    if (stats && lane == 0) {  // statistics: [4] records written by the coarse pass
        atomicAdd(fb_count + 4, (int)total);
    }
    if (total > (unsigned)rcap || m <= 0) {  // record overflow: the all-pairs kernel decides
        if (lane == 0) {
            cand_cnt[qi] = -1;
            const int slot = atomicAdd(fb_count, 1);
            fb_list[slot] = (int)qi;
        }
        return;
    }
    const unsigned thr = __float_as_uint(__uint_as_float(qmax[qi]) - window);
    const uint2* myrec = rec + (size_t)qi * rcap;
    unsigned* lc = l_cand[wave];
    int ncand = 0;  // wave-uniform
    for (unsigned e0 = 0; e0 < total; e0 += 64) {
        const unsigned e = e0 + lane;
        uint2 r = make_uint2(0u, 0u);
        if (e < total) r = myrec[e];
        const bool in = e < total && r.y >= thr;
        const unsigned long long bal = __ballot(in);
        if (in) lc[ncand + __popcll(bal & ((1ull << lane) - 1ull))] = r.x;
        ncand += __popcll(bal);
    }
    __builtin_amdgcn_wave_barrier();
    if (stats && lane == 0) {  // statistics: [2] candidate rows, [8 + b] queries with 2^(b-1) < rows <= 2^b
        atomicAdd(fb_count + 2, ncand);
        int bin = 0;
        while ((1 << bin) < ncand && bin < 15) ++bin;
        atomicAdd(fb_count + 8 + bin, 1);
    }
    if (ncand < REFINE_MIN) {
        if (lane < ncand) {
            const unsigned row = lc[lane];
            mycand[lane] = ((row / CHUNK_ROWS) << 8) | (row % CHUNK_ROWS);
        }
        if (lane == 0) cand_cnt[qi] = ncand;
        return;
    }
    RefineWave<12> R;
    R.init(q, iq, qi, b, invb, d, w2, l_row[wave], l_sc[wave]);
    for (int e0 = 0; e0 < ncand; e0 += 4) {
        const long long row = (e0 + R.g < ncand) ? (long long)lc[e0 + R.g] : -1;
        R.consider(row, R.score4(row));
    }
    if (R.overflow && ncand <= cap) {  // > REFINE_KEEP rows tie within the fp32 margin (exact duplicates): hand ALL
        for (int e = lane; e < ncand; e += 64) {  // candidates to the fp64 decision instead of the all-pairs kernel
            const unsigned row = lc[e];
            mycand[e] = ((row / CHUNK_ROWS) << 8) | (row % CHUNK_ROWS);
        }
        if (lane == 0) cand_cnt[qi] = ncand;
        return;
    }
    R.finish(qi, mycand, cand_cnt, fb_count, fb_list, stats);
}

// exact score of normalised rows, sequential k, fp64 (products of two fp32 are exact in fp64)
__device__ __forceinline__ double dot_norm_f64(const float* __restrict__ qrow_n, Rows b, int64_t brow, float invb, int d) {
    double acc = 0.0;
    for (int k = 0; k < d; k += 4) {
        const float4 bv = b.ld4(brow + k);
        const float b0 = bv.x * invb, b1 = bv.y * invb, b2 = bv.z * invb, b3 = bv.w * invb;
        acc = acc + (double)qrow_n[k + 0] * (double)b0;
        acc = acc + (double)qrow_n[k + 1] * (double)b1;
        acc = acc + (double)qrow_n[k + 2] * (double)b2;
        acc = acc + (double)qrow_n[k + 3] * (double)b3;
    }
    return acc;
}


// Exact decision among the candidates: one workgroup (4 waves) owns 64 queries; thread t < 64 = query t.
//   single-row candidates (the common case, ~1.3 per query): the block's (query, candidate) pairs are
//   flattened and taken 64 at a time; per batch and per 96-wide k chunk the four waves compute the fp64
//   products of the 64 pairs k-parallel (coalesced row segments, 16 pairs per wave; fp32 normalisation as
//   faiss leaves it, products of two fp32 are exact in fp64) into LDS, then wave 0 adds every pair's 96
//   products in ascending k -- 64 different in-order chains at once instead of one chain per wavefront.
//   whole-chunk candidates (rare): per flagged query, the lanes of wave 0 score their own rows of the chunk.
// The accumulation order is the oracle's (sequential k), ties -> lowest index, sim = (float)score.
constexpr int RS_KC = 96;             // k values per chunk (24 float4 per row)
constexpr int RS_STRIDE = RS_KC + 1;  // doubles per LDS row: 194 words == 2 (mod 64) -> conflict-free ds_read_b64
constexpr int RS_PAIRS = 1024;        // pair slots per epoch (a block has ~80 pairs; more run in further epochs)
__global__ __launch_bounds__(256) void match_rescore_kernel(Rows q, const float* __restrict__ invq,
                                                            Rows b, const float* __restrict__ invb,
                                                            int64_t n, int64_t m, int d, const int* __restrict__ cand_cnt,
                                                            const unsigned* __restrict__ cand, int cap,
                                                            int64_t* __restrict__ idx_out, float* __restrict__ sim_out, float min_sim) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* P = reinterpret_cast<double*>(smem);               // [64][RS_STRIDE] products
    double* pscore = P + 64 * RS_STRIDE;                       // [RS_PAIRS] exact score per pair of the epoch
    float* qn = reinterpret_cast<float*>(pscore + RS_PAIRS);   // [d] normalised query row (chunk rescans)
    __shared__ unsigned p_j[RS_PAIRS];                         // pair -> map row
    __shared__ unsigned char p_q[RS_PAIRS];                    // pair -> query lane
    __shared__ long long s_j[64];
    __shared__ float s_iq[64], s_ib[64];
    __shared__ int s_ql[64];
    __shared__ int s_total;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int64_t q0 = (int64_t)blockIdx.x * 64;
    // per-query state lives in wave 0 (lane = query)
    const int64_t qi = q0 + lane;
    const bool owner = wave == 0;
    const bool have = owner && qi < n;
    const float iq = have ? invq[qi] : 0.0f;
    int cnt = have ? cand_cnt[qi] : 0;
    if (iq == 0.0f || cnt < 0) cnt = 0;  // zero query: decided below; overflow (-1): match_exact_kernel's
    bool any_rescan = false;
    int my_off = 0, my_pairs = 0;
    const unsigned* mycand = cand + (size_t)(have ? qi : 0) * cap;
    auto row_of = [&](unsigned ce) { return (long long)(ce >> 8) * CHUNK_ROWS + (ce & 127u); };
    if (owner) {
        // flatten: the single-row candidates of query `lane` become pairs [my_off, my_off + my_pairs)
        for (int e = 0; e < cnt; ++e) {
            const unsigned ce = mycand[e];
            if (ce & 128u) any_rescan = true;
            else if (row_of(ce) < m) ++my_pairs;
        }
        int incl = my_pairs;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        my_off = incl - my_pairs;
        if (lane == 63) s_total = incl;
    }
    __syncthreads();
    const int total = s_total;
    double best = 0.0;
    long long bj = -1;
    for (int E0 = 0; E0 < total; E0 += RS_PAIRS) {  // one epoch unless a block has > RS_PAIRS pairs
        const int ecount = min(RS_PAIRS, total - E0);
        if (owner) {
            int k = my_off;
            for (int e = 0; e < cnt; ++e) {
                const unsigned ce = mycand[e];
                if ((ce & 128u) || row_of(ce) >= m) continue;
                if (k >= E0 && k < E0 + RS_PAIRS) {
                    p_j[k - E0] = (unsigned)row_of(ce);
                    p_q[k - E0] = (unsigned char)lane;
                }
                ++k;
            }
        }
        __syncthreads();
        for (int p0 = 0; p0 < ecount; p0 += 64) {
            // slot `lane` of this batch = pair p0 + lane of the epoch
            long long j = -1;
            if (owner) {
                const int pid = p0 + lane;
                const int ql = (pid < ecount) ? (int)p_q[pid] : 0;
                const float iqq = __shfl(iq, ql);  // all lanes of wave 0 take part
                float ibb = 0.f;
                if (pid < ecount) {
                    j = (long long)p_j[pid];
                    ibb = invb[j];
                }
                s_j[lane] = j;
                s_ql[lane] = ql;
                s_iq[lane] = iqq;
                s_ib[lane] = ibb;
            }
            __syncthreads();
            double acc = 0.0;
            // wave w takes slots 16 w .. 16 w + 15, two per pass (lanes 0..23 and 32..55: one float4 of k each);
            // the row segments of chunk c+1 are fetched while wave 0 runs the chains of chunk c
            const int sub = lane >> 5, l4 = lane & 31;
            float4 qv[8], bv[8];
            auto fetch = [&](int k0) {
                const int kn = min(RS_KC, d - k0);
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int sl = 16 * wave + 2 * it + sub;
                    const long long jj = s_j[sl];
                    if (jj >= 0 && 4 * l4 < kn) {
                        qv[it] = q.ld4((q0 + s_ql[sl]) * (int64_t)d + k0 + 4 * l4);
                        bv[it] = b.ld4(jj * (int64_t)d + k0 + 4 * l4);
                    }
                }
            };
            fetch(0);
            for (int k0 = 0; k0 < d; k0 += RS_KC) {
                const int kn = min(RS_KC, d - k0);  // d % 4 == 0
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int sl = 16 * wave + 2 * it + sub;
                    if (s_j[sl] >= 0 && 4 * l4 < kn) {
                        const float iqq = s_iq[sl], ibb = s_ib[sl];
                        double* dst = P + sl * RS_STRIDE + 4 * l4;
                        dst[0] = (double)(qv[it].x * iqq) * (double)(bv[it].x * ibb);
                        dst[1] = (double)(qv[it].y * iqq) * (double)(bv[it].y * ibb);
                        dst[2] = (double)(qv[it].z * iqq) * (double)(bv[it].z * ibb);
                        dst[3] = (double)(qv[it].w * iqq) * (double)(bv[it].w * ibb);
                    }
                }
                __syncthreads();
                if (k0 + RS_KC < d) fetch(k0 + RS_KC);
                if (owner && j >= 0) {
                    const double* src = P + lane * RS_STRIDE;
                    int k = 0;
                    for (; k + 8 <= kn; k += 8) {  // reads first, then the in-order chain
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = src[k + u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc = acc + v[u];
                    }
                    for (; k < kn; ++k) acc = acc + src[k];
                }
                __syncthreads();
            }
            if (owner && j >= 0) pscore[p0 + lane] = acc;
        }
        __syncthreads();
        if (owner) {  // every query folds its own pairs of this epoch: best score, ties -> lowest index
            const int lo = max(my_off, E0), hi = min(my_off + my_pairs, E0 + RS_PAIRS);
            for (int k = lo; k < hi; ++k) {
                const long long j = (long long)p_j[k - E0];
                const double sc = pscore[k - E0];
                if (bj < 0 || sc > best || (sc == best && j < bj)) {
                    best = sc;
                    bj = j;
                }
            }
        }
        __syncthreads();
    }
    if (!owner) return;
    // whole-chunk candidates: wave 0 takes the flagged queries one by one
    if (__any(any_rescan)) {
        for (int ql = 0; ql < 64; ++ql) {
            if (!__shfl((int)any_rescan, ql)) continue;  // wave-uniform
            const int64_t qq = q0 + ql;
            const float iqq = __shfl(iq, ql);
            const int cq = __shfl(cnt, ql);
            __builtin_amdgcn_wave_barrier();
            for (int k = lane * 4; k < d; k += 256) {
                float4 v = q.ld4(qq * (int64_t)d + k);
                v.x = v.x * iqq; v.y = v.y * iqq; v.z = v.z * iqq; v.w = v.w * iqq;
                *reinterpret_cast<float4*>(qn + k) = v;
            }
            __builtin_amdgcn_wave_barrier();
            double rbest = 0.0;
            long long rj = -1;
            for (int e = 0; e < cq; ++e) {
                const unsigned ce = cand[(size_t)qq * cap + e];
                if (!(ce & 128u)) continue;
                const long long base = (long long)(ce >> 8) * CHUNK_ROWS;
                for (int li = lane; li < CHUNK_ROWS; li += 64) {
                    const long long j = base + li;
                    if (j < m) {
                        const double sc = dot_norm_f64(qn, b, j * (int64_t)d, invb[j], d);
                        if (rj < 0 || sc > rbest || (sc == rbest && j < rj)) {
                            rbest = sc;
                            rj = j;
                        }
                    }
                }
            }
            wave_argmax(rbest, rj);
            if (lane == ql && rj >= 0 && (bj < 0 || rbest > best || (rbest == best && rj < bj))) {
                best = rbest;
                bj = rj;
            }
        }
    }
    if (!have) return;
    if (iq == 0.0f) {  // zero query: every score is 0.0, the lowest index wins
        idx_out[qi] = (m > 0) ? 0 : -1;
        sim_out[qi] = 0.0f;
    } else if (cand_cnt[qi] >= 0) {
        // (half-width pass: min_sim = the gate -- the lists hold only rows of chunks that could reach it, so a best below it
        // is not the oracle's arg-max but the proof that the query has no match; otherwise min_sim = -Inf)
        const bool ok = !((float)best < min_sim) && bj >= 0;
        idx_out[qi] = ok ? bj : -1;
        sim_out[qi] = ok ? (float)best : -2.0f;
    } else if (cand_cnt[qi] == -2) {  // below the caller's gate (match_select_kernel)
        idx_out[qi] = -1;
        sim_out[qi] = -2.0f;
    }
}

// all-pairs exact decision for the queries in `list` (or all queries if list == NULL)
__global__ __launch_bounds__(256) void match_exact_kernel(Rows q, const float* __restrict__ invq,
                                                          Rows b, const float* __restrict__ invb,
                                                          int64_t n, int64_t m, int d, const int* __restrict__ list,
                                                          const int* __restrict__ list_count,
                                                          int64_t* __restrict__ idx_out, float* __restrict__ sim_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* qn = reinterpret_cast<float*>(smem);
    double* rs = reinterpret_cast<double*>(smem + (((size_t)d * 4 + 15) & ~(size_t)15));
    long long* rj = reinterpret_cast<long long*>(rs + 4);
    const int64_t count = list ? (int64_t)*list_count : n;
    for (int64_t e = blockIdx.x; e < count; e += gridDim.x) {
        const int64_t qi = list ? (int64_t)list[e] : e;
        const float iq = invq ? invq[qi] : 1.0f;
        __syncthreads();
        for (int k = threadIdx.x; k < d; k += 256) qn[k] = q.ld1(qi * (int64_t)d + k) * iq;
        __syncthreads();
        double best = 0.0;
        long long bj = -1;
        for (long long j = threadIdx.x; j < m; j += 256) {
            const double s = dot_norm_f64(qn, b, j * (int64_t)d, invb ? invb[j] : 1.0f, d);
            if (bj < 0 || s > best) {
                best = s;
                bj = j;
            }
        }
        wave_argmax(best, bj);
        if (lane_id() == 0) {
            rs[threadIdx.x >> 6] = best;
            rj[threadIdx.x >> 6] = bj;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (rj[w] >= 0 && (rj[0] < 0 || rs[w] > rs[0] || (rs[w] == rs[0] && rj[w] < rj[0]))) {
                    rs[0] = rs[w];
                    rj[0] = rj[w];
                }
            idx_out[qi] = (m > 0) ? rj[0] : -1;
            sim_out[qi] = (m > 0) ? (float)rs[0] : 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// threshold + stable compaction (VoxelHashMap.cpp:501-511, 587-600), single workgroup
// ---------------------------------------------------------------------------------------------
// (one workgroup: the order of the survivors is the order of the queries.  A pass takes TC_ITERS x 1024 queries: all of a thread's
// similarities are loaded at once, the per-wave counts of every 1024-query slab go to the LDS behind ONE barrier, and the gathers
// of the survivors -- matched row, then its coordinates -- are in flight together; a slab at a time, three dependent round trips and
// three barriers each, it was 25 us for 20 000 queries on the path every registration's RANSAC waits for.)
constexpr int TC_ITERS = 4;   // (8: the arrays spilled at 1024 threads)
__global__ __launch_bounds__(1024) void threshold_compact_kernel(const float* __restrict__ sim, const int64_t* __restrict__ idx,
                                                                 int64_t n, double thr, int64_t* __restrict__ keep,
                                                                 int64_t* __restrict__ count, int32_t* __restrict__ corres,
                                                                 const double* __restrict__ qxyz, const double* __restrict__ bxyz,
                                                                 double* __restrict__ src_out, double* __restrict__ tgt_out) {
    __shared__ int wsum[TC_ITERS][16];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    int64_t base = 0;   // survivors in front of this pass (the same value in every thread)
    for (int64_t s0 = 0; s0 < n; s0 += (int64_t)TC_ITERS * 1024) {
        bool valid[TC_ITERS];
        int before[TC_ITERS];
        float sv[TC_ITERS];
#pragma unroll
        for (int t = 0; t < TC_ITERS; ++t) {
            const int64_t i = s0 + (int64_t)t * 1024 + threadIdx.x;
            sv[t] = i < n ? sim[i] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < TC_ITERS; ++t) {
            const int64_t i = s0 + (int64_t)t * 1024 + threadIdx.x;
            valid[t] = (i < n) && !((double)sv[t] < thr);
            const unsigned long long bal = __ballot(valid[t]);
            before[t] = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wsum[t][wave] = __popcll(bal);
        }
        __syncthreads();
        int64_t k[TC_ITERS];
        int j[TC_ITERS];   // (map rows fit 31 bits: the candidate entries hold chunk << 8)
        int64_t run = base;
#pragma unroll
        for (int t = 0; t < TC_ITERS; ++t) {
            int woff = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const int c = wsum[t][w];
                if (w < wave) woff += c;
                tot += c;
            }
            k[t] = run + woff + before[t];
            run += tot;
            const int64_t i = s0 + (int64_t)t * 1024 + threadIdx.x;
            j[t] = (valid[t] && idx) ? (int)idx[i] : 0;
        }
        base = run;
#pragma unroll
        for (int t = 0; t < TC_ITERS; ++t) {
            if (!valid[t]) continue;
            const int64_t i = s0 + (int64_t)t * 1024 + threadIdx.x;
            keep[k[t]] = i;
            if (corres) {
                corres[2 * k[t] + 0] = (int32_t)i;
                corres[2 * k[t] + 1] = (int32_t)j[t];
            }
            if (src_out) {
                src_out[3 * k[t] + 0] = qxyz[3 * i + 0];
                src_out[3 * k[t] + 1] = qxyz[3 * i + 1];
                src_out[3 * k[t] + 2] = qxyz[3 * i + 2];
            }
            if (tgt_out) {
                tgt_out[3 * k[t] + 0] = bxyz[3 * (int64_t)j[t] + 0];
                tgt_out[3 * k[t] + 1] = bxyz[3 * (int64_t)j[t] + 1];
                tgt_out[3 * k[t] + 2] = bxyz[3 * (int64_t)j[t] + 2];
            }
        }
        __syncthreads();   // wsum is rewritten by the next pass
    }
    if (threadIdx.x == 0) *count = base;
}


// 1/|row| only (EXACT mode)
__global__ __launch_bounds__(256) void inv_norm_kernel(const float* __restrict__ x, int64_t rows, int d,
                                                       float* __restrict__ inv_out) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float4 v[4];
    float nr = row_sumsq_wave(x + r * (int64_t)d, d, v);
    float inv = inv_norm_from_sumsq(nr);
    // faiss leaves zero rows untouched: scaling by 1 reproduces that
    if (lane_id() == 0) inv_out[r] = (nr > 0.0f) ? inv : 1.0f;
}

}  // namespace

// stage 2 of a search: candidate selection + exact fp64 decision (reads ws of stage 1)
// gated: the search was started by the gated family (do_search_coarse(..., gated)); gate: queries whose best similarity is
// provably below it are reported as (-1, -2.0) instead of being resolved (int8 pass only; -Inf = resolve every query)
int do_search_finish(Rows q, const void* qprep, int64_t n, Rows b, const void* bprep, int64_t m, int d,
                     int64_t* idx_out, float* sim_out, void* ws, hipStream_t st, bool gated, float gate, int records) {
    Prepared Q = carve_prepared(const_cast<void*>(qprep), n, d);
    Prepared B = carve_prepared(const_cast<void*>(bprep), m, d);
    SearchWs w = carve_search(ws, n, m);
    const CoarseArgs a = coarse_args(Q, B, w, n, m, coarse_qblock(d));
    const float w2 = 2.0f * (float)(d / 16 + 4 + 2) * 5.9604645e-8f;
    const bool i8 = records != VFM_RECORDS_F16 && use_i8(d, n, m, gated);
    if (i8 && !gated) records = VFM_RECORDS_TOP2;  // as do_search_coarse chose
    records = effective_records(records, d, n, m);
    // the fp6 coarse kernel has left the survivors in its workgroups' slots (half width, or -- VFM_RECORDS_MX6_FUSED -- full width: behind
    // the kernel the two are the same search: survivors of a gate test, binned, rescanned on the int8 image with the gate as hit test)
    const bool fused6 = i8 && (records == VFM_RECORDS_MX6_HALF_FUSED || records == VFM_RECORDS_MX6_FUSED);
    const bool fused = i8 && (records == VFM_RECORDS_HALF_FUSED || fused6);   // the coarse kernel has done the selection already
    const bool mx6half = i8 && records == VFM_RECORDS_MX6_HALF;   // the half-width pass on the fp6 image: its bounds in the selection
    const bool half = i8 && (records == VFM_RECORDS_HALF || fused || mx6half);
    if (mx6half || fused6) records = VFM_RECORDS_HALF;
    if (half && !(gate > -__builtin_inff())) return vfm_fail(VFM_EINVAL, "search_finish: VFM_RECORDS_HALF needs a finite gate");
    if (!i8 && use_sparse(d, n, m)) {
        hipLaunchKernelGGL(match_filter_refine_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, q, Q.inv, b, B.inv, n, m, d,
                           DEFAULT_WINDOW, w2, w.qmax, w.rec_cnt, w.rec, w.rcap, w.cand_cnt, w.cand, w.cap, w.fb_count, w.fb_list, vfm_cfg().match_stats);
        VFM_CHECK_LAUNCH("match_filter_refine_kernel");
    } else {
        const int chunk_lds = i8 && (size_t)a.nchunks * sizeof(float2) <= 63 * 1024;  // (step, max E) of every chunk in LDS
        bool use_bins = false;
        // best-score records with many queries per map chunk: the rescan runs chunk-major (match_rescan_chunk_kernel)
        // records of the fp6 pass: its own bounds in the selection; behind it the int8 image and bounds, as for the int8 kinds
        const bool pilot = i8 && records == VFM_RECORDS_MX6_PILOT;
        if (pilot) records = VFM_RECORDS_MX6;
        const bool mx6 = i8 && (records == VFM_RECORDS_MX6 || records == VFM_RECORDS_MX6_TOP2);
        const bool top2 = i8 && (records == VFM_RECORDS_TOP2 || records == VFM_RECORDS_MX6_TOP2);
        if (mx6) records = top2 ? VFM_RECORDS_TOP2 : VFM_RECORDS_BEST;
        const bool best = i8 && records == VFM_RECORDS_BEST && vfm_cfg().select_variant != 1;
        use_bins = (best || half) && vfm_cfg().select_variant != 2 && n >= 4 * (int64_t)a.nchunks;
        if (pilot && use_bins) {
            // the pilot rescan: one chunk per query, scored exactly on the int8 image, raises qmax (match_rescan_chunk_kernel, pilot
            // branch); the bins it used are emptied again for the selection
            hipLaunchKernelGGL(match_pilot_bin_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, (const unsigned long long*)w.qbest,
                               (const float*)Q.inv, w.bin_cnt, w.bins, w.bin_cap, a.nchunks);
            VFM_CHECK_LAUNCH("match_pilot_bin_kernel");
            const int rc = launch_rescan_chunk(w, a.nchunks, n, m, d, i8_bounds(Q, B, true, VFM_RECORDS_BEST), Q, B, 0, gate, (const int*)nullptr,
                                               L2Terms{nullptr, nullptr, 0.0f}, st, true);
            if (rc != VFM_OK) return rc;
            VFM_CHECK_HIP(hipMemsetAsync(w.bin_cnt, 0, (size_t)((a.nchunks + 63) / 64 * 64) * BIN_CNT_STRIDE * sizeof(unsigned), st));
        }
        if (fused6) {
            hipLaunchKernelGGL(match_bin_survivors_kernel, dim3(256), dim3(256), 0, st, reinterpret_cast<const unsigned*>(w.partials),
                               mx6_survivor_slot_words(), (const int*)w.fb_count, w.bin_cnt, w.bins, w.bin_cap, w.cand_cnt, w.cand, w.cap);
        } else if (fused) {
            // (nothing to select)
        } else if (half) {
            const int half_lds = (size_t)a.nchunks * 12 <= 63 * 1024;  // (step, max E, max |rest|) of every chunk in LDS
            hipLaunchKernelGGL(match_select_half_kernel, dim3((unsigned)a.nq_tiles), dim3(64 * 8), half_lds ? (size_t)a.nchunks * 12 : 0,
                               st, reinterpret_cast<const unsigned*>(w.partials), a.nchunks, n, Q.inv,
                               mx6half ? mx6_bounds_half(Q, B) : i8_bounds(Q, B, true, records), (const float*)Q.rest, (const float*)B.grest, gate, half_lds, w.cand_cnt, w.cand, w.cap, w.fb_count,
                               w.fb_list, vfm_cfg().match_stats, use_bins ? w.bin_cnt : (unsigned*)nullptr, use_bins ? w.bins : (int*)nullptr,
                               w.bin_cap);
        } else if (top2 && vfm_cfg().select_variant != 1) {
            hipLaunchKernelGGL(match_select_top2_kernel, dim3((unsigned)a.nq_tiles), dim3(64 * SELECT_TOP2_WAVES),
                               chunk_lds ? (size_t)a.nchunks * sizeof(float2) : 0, st, (const uint2*)w.partials, a.nchunks, n,
                               a.first_pad_chunk, (const unsigned*)w.qmax, Q.inv, mx6 ? mx6_bounds(Q, B, 1) : i8_bounds(Q, B, true, records), gate, chunk_lds,
                               w.cand_cnt, w.cand, w.cap, w.fb_count, w.fb_list, vfm_cfg().match_stats);
        } else if (best) {
            // (chunk_lds 2: room for the per-chunk histogram of the tile's candidates as well -- chunks must fit 16 bits of a staged entry)
            const int best_lds = chunk_lds && use_bins && (size_t)a.nchunks * 12 <= 63 * 1024 && a.nchunks < 65536 ? 2 : chunk_lds;
            hipLaunchKernelGGL(match_select_best_kernel, dim3((unsigned)a.nq_tiles), dim3(64 * SELECT_BEST_WAVES),
                               best_lds == 2 ? (size_t)a.nchunks * 12 : chunk_lds ? (size_t)a.nchunks * sizeof(float2) : 0, st, reinterpret_cast<const unsigned*>(w.partials),
                               a.nchunks, n, (const unsigned*)w.qmax, Q.inv, mx6 ? mx6_bounds(Q, B) : i8_bounds(Q, B, true, records), gate,
                               best_lds, w.cand_cnt,
                               w.cand, w.cap, w.fb_count, w.fb_list, vfm_cfg().match_stats, use_bins ? w.bin_cnt : (unsigned*)nullptr,
                               use_bins ? w.bins : (int*)nullptr, a.first_pad_chunk, w.bin_cap);
        } else
        hipLaunchKernelGGL(match_select_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * SELECT_GROUPS),
                           chunk_lds ? (size_t)a.nchunks * sizeof(float2) : 0, st, w.partials, a.nchunks, a.npad, n, a.first_pad_chunk, w.qmax,
                           Q.inv, DEFAULT_WINDOW, mx6 ? mx6_bounds(Q, B, top2 ? 1 : 0) : i8_bounds(Q, B, i8, records), gate, chunk_lds, w.cand_cnt, w.cand, w.cap,
                           w.fb_count, w.fb_list, vfm_cfg().match_stats);
        VFM_CHECK_LAUNCH("match_select_kernel");
        if (i8) {  // candidate chunks -> candidate rows (the record buffer of the fp16 pass is free: it holds the hit lists)
            const int* guard = half ? w.fb_count + HALF_GUARD_FLAG : (const int*)nullptr;
            if (half) {   // the device-side guard of the half-width pass (see half_guard_kernel)
                hipLaunchKernelGGL(half_guard_kernel, dim3(1), dim3(256), 0, st, w.fb_count, (const unsigned*)w.bin_cnt, a.nchunks,
                                   fused ? 1 : 0, (long long)HALF_GUARD_PER_QUERY * (long long)n);
                VFM_CHECK_LAUNCH("half_guard_kernel");
            }
            hipLaunchKernelGGL(match_rescan_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, m, d, i8_bounds(Q, B, true, records),
                               (const uint4*)Q.tiles8, (const uint4*)B.tiles8, (const unsigned*)w.qmax, w.cand_cnt, w.cand, w.cap,
                               reinterpret_cast<unsigned*>(w.rec), 2 * w.rcap, w.fb_count, w.fb_list, half ? 1 : 0, gate, guard,
                               L2Terms{nullptr, nullptr, 0.0f});
            VFM_CHECK_LAUNCH("match_rescan_kernel");
            if (half) {
#define VFM_GATEPASS(KS)                                                                                                        \
    hipLaunchKernelGGL(match_gatepass_kernel<KS>, dim3((unsigned)a.nchunks), dim3(256), 0, st, n, m, a.nq_tiles,                  \
                       i8_bounds(Q, B, true, records), (const float*)Q.inv, (const uint4*)Q.tiles8, (const uint4*)B.tiles8, gate, \
                       w.cand_cnt, w.cand, w.cap, guard, w.qmax)
                switch (d / 32) {
                    case 8: VFM_GATEPASS(8); break;
                    case 12: VFM_GATEPASS(12); break;
                    case 16: VFM_GATEPASS(16); break;
                    case 20: VFM_GATEPASS(20); break;
                    default: VFM_GATEPASS(24); break;
                }
#undef VFM_GATEPASS
                VFM_CHECK_LAUNCH("match_gatepass_kernel");
            }
            if (use_bins) {
                const int rc = launch_rescan_chunk(w, a.nchunks, n, m, d, i8_bounds(Q, B, true, records), Q, B, half ? 1 : 0, gate, guard,
                                                   L2Terms{nullptr, nullptr, 0.0f}, st);
                if (rc != VFM_OK) return rc;
            }
            hipLaunchKernelGGL(match_rescan_close_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, w.cand_cnt, w.cap,
                               w.fb_count, w.fb_list, reinterpret_cast<int*>(w.rec_cnt), half ? (const float*)Q.inv : (const float*)nullptr,
                               (const unsigned*)w.hit_cnt);
            VFM_CHECK_LAUNCH("match_rescan_close_kernel");
            // (rec_cnt, unused by the int8 pass, holds the list of crowded queries; fb_count[6] its length)
            // (round 4: one wave per possible entry after all.  With 1024 workgroups striding over the list a wave had two or three
            // queries of 30 - 40 us each and the kernel lasted as long as the unluckiest wave -- 137 us at 9891 crowded queries; a wave per
            // entry is placed as slots come free: 125 us.  Handing the entries out through an atomic cursor was tried: 186 us -- 4096
            // waves ask the same address at once and every later load of a wave waits behind its atomic.  The refinement is 124 registers
            // since R4.5, and workgroups past the list's end return at once: ~2 us for 5000 of them.)
            const unsigned grid = (unsigned)((n + 3) / 4);
#define VFM_REFINE(NT)                                                                                                          \
    hipLaunchKernelGGL(match_refine_kernel<NT>, dim3(grid), dim3(256), 0, st, q, Q.inv, b, B.inv, n, m, d, w2, w.cand_cnt, w.cand, \
                       w.cap, w.fb_count, w.fb_list, vfm_cfg().match_stats, (const int*)w.rec_cnt, (const int*)(w.fb_count + 6),            \
                       use_bins ? (const float*)w.cand_up : (const float*)nullptr, (const unsigned*)w.hit_cnt,                     \
                       i8_bounds(Q, B, true, records))
            if (d <= 256) VFM_REFINE(4);
            else if (d <= 384) VFM_REFINE(6);
            else if (d <= 512) VFM_REFINE(8);
            else VFM_REFINE(12);
#undef VFM_REFINE
        } else {
            hipLaunchKernelGGL(match_refine_kernel<12>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, q, Q.inv, b, B.inv, n, m, d, w2,
                               w.cand_cnt, w.cand, w.cap, w.fb_count, w.fb_list, vfm_cfg().match_stats, (const int*)nullptr, (const int*)nullptr,
                               (const float*)nullptr, (const unsigned*)nullptr, I8Bounds{});
        }
        VFM_CHECK_LAUNCH("match_refine_kernel");
    }
    {
        const size_t lds = (size_t)(64 * RS_STRIDE + RS_PAIRS) * sizeof(double) + (size_t)d * sizeof(float);
        static unsigned long long attr_set = 0ull;  // one bit per device
        if (!attr_done(attr_set)) {
            VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_rescore_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            attr_mark(attr_set);
        }
        hipLaunchKernelGGL(match_rescore_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), lds, st, q, Q.inv, b, B.inv, n, m,
                           d, w.cand_cnt, w.cand, w.cap, idx_out, sim_out, half ? gate : -__builtin_inff());
    }
    VFM_CHECK_LAUNCH("match_rescore_kernel");
    hipLaunchKernelGGL(match_exact_kernel, dim3(256), dim3(256), (((size_t)d * 4 + 15) & ~(size_t)15) + 64, st, q, Q.inv,
                       b, B.inv, n, m, d, w.fb_list, w.fb_count, idx_out, sim_out);
    VFM_CHECK_LAUNCH("match_exact_kernel(fallback)");
    return VFM_OK;
}

int launch_i8_rescans(const SearchWs& w, const CoarseArgs& a, const Prepared& Q, const Prepared& B, int64_t n, int64_t m, int d,
                      bool use_bins, L2Terms l2, hipStream_t st) {
    const I8Bounds ib = i8_bounds(Q, B, true, VFM_RECORDS_BEST);
    hipLaunchKernelGGL(match_rescan_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, m, d, ib, (const uint4*)Q.tiles8,
                       (const uint4*)B.tiles8, (const unsigned*)w.qmax, w.cand_cnt, w.cand, w.cap, reinterpret_cast<unsigned*>(w.rec),
                       2 * w.rcap, w.fb_count, w.fb_list, 0, 0.0f, (const int*)nullptr, l2);
    VFM_CHECK_LAUNCH("match_rescan_kernel");
    if (use_bins) {
        const int rc = launch_rescan_chunk(w, a.nchunks, n, m, d, ib, Q, B, 0, 0.0f, (const int*)nullptr, l2, st);
        if (rc != VFM_OK) return rc;
    }
    hipLaunchKernelGGL(match_rescan_close_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, w.cand_cnt, w.cap, w.fb_count,
                       w.fb_list, reinterpret_cast<int*>(w.rec_cnt), (const float*)nullptr, (const unsigned*)w.hit_cnt);
    VFM_CHECK_LAUNCH("match_rescan_close_kernel");
    return VFM_OK;
}

// vfm_match_search_probe_half: how many (query, chunk) pairs the half-width pass would leave for the rescan -- its coarse
// pass has just run into ws (do_search_coarse(..., VFM_RECORDS_HALF)); only fb_count[5] is written
int probe_half_select(const void* qprep, int64_t n, const void* bprep, int64_t m, int d, void* ws, float gate, hipStream_t st) {
    Prepared Q = carve_prepared(const_cast<void*>(qprep), n, d);
    Prepared B = carve_prepared(const_cast<void*>(bprep), m, d);
    SearchWs w = carve_search(ws, n, m);
    const CoarseArgs a = coarse_args(Q, B, w, n, m, coarse_qblock(d));
    const int half_lds = (size_t)a.nchunks * 12 <= 63 * 1024;
    hipLaunchKernelGGL(match_select_half_kernel, dim3((unsigned)a.nq_tiles), dim3(64 * 8), half_lds ? (size_t)a.nchunks * 12 : 0, st,
                       reinterpret_cast<const unsigned*>(w.partials), a.nchunks, n, Q.inv, i8_bounds(Q, B, true, VFM_RECORDS_HALF),
                       (const float*)Q.rest, (const float*)B.grest, gate, half_lds, w.cand_cnt, (unsigned*)nullptr, w.cap, w.fb_count,
                       w.fb_list, 0, (unsigned*)nullptr, (int*)nullptr, w.bin_cap);
    VFM_CHECK_LAUNCH("match_select_half_kernel(probe)");
    return VFM_OK;
}

// dense fp16 records -> candidate lists (the Euclidean search's use of match_select_kernel)
int launch_select_dense(const SearchWs& w, const CoarseArgs& a, const float* qinv, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(match_select_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * SELECT_GROUPS), 0, st, w.partials, a.nchunks,
                       a.npad, n, a.first_pad_chunk, w.qmax, qinv, DEFAULT_WINDOW, i8_bounds(Prepared{}, Prepared{}, false),
                       -__builtin_inff(), 0, w.cand_cnt, w.cand, w.cap, w.fb_count, w.fb_list, vfm_cfg().match_stats);
    VFM_CHECK_LAUNCH("match_select_kernel");
    return VFM_OK;
}

// EXACT mode of vfm_match_ip_top1: all-pairs fp64; ws holds 1/|row| of both operands
int exact_ip_top1(const float* q, int64_t n, const float* b, int64_t m, int d, int64_t* idx_out, float* sim_out, void* ws,
                  hipStream_t st) {
    VfmCarver c(ws);
    float* invq = c.take<float>((size_t)n);
    float* invb = c.take<float>((size_t)m);
    hipLaunchKernelGGL(inv_norm_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, q, n, d, invq);
    hipLaunchKernelGGL(inv_norm_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, b, m, d, invb);
    VFM_CHECK_LAUNCH("inv_norm_kernel");
    const unsigned grid = (unsigned)(n < 4096 ? n : 4096);
    hipLaunchKernelGGL(match_exact_kernel, dim3(grid), dim3(256), (((size_t)d * 4 + 15) & ~(size_t)15) + 64, st, q,
                       invq, b, invb, n, m, d, (const int*)nullptr, (const int*)nullptr, idx_out, sim_out);
    VFM_CHECK_LAUNCH("match_exact_kernel");
    return VFM_OK;
}

}  // namespace vfmm

using namespace vfmm;

VFM_EXPORT int vfm_threshold_compact(const float* sim, const int64_t* idx, int64_t n, double thr, int64_t* keep_out,
                                     int64_t* count_out, int32_t* corres_out, const double* q_xyz, const double* b_xyz,
                                     double* src_xyz_out, double* tgt_xyz_out, vfm_stream_t stream) {
    VFM_CHECK_ARG(n >= 0 && sim && keep_out && count_out, "threshold_compact: bad arguments");
    VFM_CHECK_ARG(!(corres_out || tgt_xyz_out) || idx, "threshold_compact: idx required for corres / tgt output");
    VFM_CHECK_ARG(!src_xyz_out || q_xyz, "threshold_compact: q_xyz required");
    VFM_CHECK_ARG(!tgt_xyz_out || b_xyz, "threshold_compact: b_xyz required");
    hipLaunchKernelGGL(threshold_compact_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sim, idx, n, thr, keep_out,
                       count_out, corres_out, q_xyz, b_xyz, src_xyz_out, tgt_xyz_out);
    VFM_CHECK_LAUNCH("threshold_compact_kernel");
    return VFM_OK;
}
