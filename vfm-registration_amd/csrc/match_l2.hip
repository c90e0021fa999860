// match_l2.hip -- exact Euclidean nearest neighbours both ways (row A6: find_correspondences, registration_node.py:482-538)
// on the fp16 coarse pass: common power-of-two scale, the norm term in two appended fp16 columns (or a row bias in the
// accumulator start), fp64 decision in the oracle's order ("Row A6").
#include "match_internal.h"

namespace vfmm {
namespace {

// ---------------------------------------------------------------------------------------------
// exact Euclidean 1-NN (find_correspondences, registration_node.py:485-496): one workgroup per
// query row, fp64 squared distance, sequential k, ties -> lowest index
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nn_l2_kernel(const float* __restrict__ a, int64_t n, const float* __restrict__ b,
                                                    int64_t m, int d, const int* __restrict__ list,
                                                    const int* __restrict__ list_count, int64_t* __restrict__ nn,
                                                    double* __restrict__ d2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* qa = reinterpret_cast<float*>(smem);
    double* rs = reinterpret_cast<double*>(smem + (((size_t)d * 4 + 15) & ~(size_t)15));
    long long* rj = reinterpret_cast<long long*>(rs + 4);
    const int64_t count = list ? (int64_t)*list_count : n;  // list: the queries whose candidate list overflowed
    for (int64_t e = blockIdx.x; e < count; e += gridDim.x) {
        const int64_t qi = list ? (int64_t)list[e] : e;
        __syncthreads();
        for (int k = threadIdx.x; k < d; k += 256) qa[k] = a[qi * (int64_t)d + k];
        __syncthreads();
        double best = 0.0;
        long long bj = -1;
        for (long long j = threadIdx.x; j < m; j += 256) {
            const float* p = b + j * (int64_t)d;
            double acc = 0.0;
            for (int k = 0; k < d; ++k) {
                const double t = (double)qa[k] - (double)p[k];
                acc = acc + t * t;
            }
            if (bj < 0 || acc < best) {
                best = acc;
                bj = j;
            }
        }
        // arg-min: negate so that wave_argmax applies (x -> -x is exact)
        double neg = -best;
        wave_argmax(neg, bj);
        if (lane_id() == 0) {
            rs[threadIdx.x >> 6] = neg;
            rj[threadIdx.x >> 6] = bj;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (rj[w] >= 0 && (rj[0] < 0 || rs[w] > rs[0] || (rs[w] == rs[0] && rj[w] < rj[0]))) {
                    rs[0] = rs[w];
                    rj[0] = rj[w];
                }
            nn[qi] = rj[0];
            if (d2) d2[qi] = -rs[0];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// FAST Euclidean 1-NN (row A6): the arg-min of |a - b|^2 = |a|^2 + |b|^2 - 2 a.b over b is the arg-max
// of f(b) = a~.b~ - |b~|^2 / 2 for any common scale (x~ = x * 2^-k, k chosen so that every row norm of
// BOTH sets is <= 1: power of two, so the scaling is exact).  f comes out of the same fp16 MFMA coarse
// pass as the cosine search by appending two columns to the fragment tiles: the map row carries
// (-hi, -lo) with hi + lo = |b~|^2 / 2 split into two fp16 values (error 2^-22), the query row
// carries (1, 1); the padded K is the next multiple of 128 >= d + 2.  |coarse - f| <= E as before
// (operand rounding (2u + u^2) |a~| |b~| <= 9.8e-4, subnormal flush < 1e-6, norm term < 1e-6), so the
// same window / select apply; the decision among the candidates is the oracle's fp64 squared
// distance (sequential k), ties -> lowest index.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_maxnorm_kernel(const float* __restrict__ x, int64_t rows, int d,
                                                         unsigned* __restrict__ max_bits) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float p = 0.f;
    for (int k = lane_id(); k < d; k += 64) {
        const float v = x[r * (int64_t)d + k];
        p += v * v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p += __shfl_xor(p, off);
    if (lane_id() == 0) atomicMax(max_bits, __float_as_uint(p));  // p >= 0: bit order == value order
}

// 2^-k with 2^k >= sqrt(max sum of squares) * 1.01 (the fp32 sums above are good to ~1e-6)
__device__ __forceinline__ float l2_scale(const unsigned* max_bits) {
    const float mx = __uint_as_float(*max_bits);
    if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.0f;
    const float s = sqrtf(mx) * 1.01f;
    int e;
    (void)frexpf(s, &e);  // s = f * 2^e, f in [0.5, 1)  =>  2^e > s
    return ldexpf(1.0f, -e);
}

// one workgroup (4 waves) per 32-row tile; role 0 = query (extra columns 1, 1), 1 = map (-hi, -lo)
// aug = 1: the norm term travels in two appended columns (d + 2 <= kp); aug = 0 (d > 510, kp = d rounded up
// to 128): no extra columns, the MAP role stores -|b~|^2 / 2 per row in inv_out instead -- the coarse kernel
// adds it to the accumulator start of that row (CoarseArgs::row_bias)
__global__ __launch_bounds__(256) void l2_prep_kernel(const float* __restrict__ x, int64_t rows, int d, int kp,
                                                      const unsigned* __restrict__ max_bits, int role, int aug,
                                                      float* __restrict__ inv_out, uint4* __restrict__ tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tile = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const float scale = l2_scale(max_bits);
    _Float16* img = reinterpret_cast<_Float16*>(smem);
    for (int pr = wave; pr < TILE_ROWS; pr += 4) {
        const int64_t r = (int64_t)tile * TILE_ROWS + pr;
        const bool valid = r < rows;
        float nb2 = 0.f;
        for (int k = lane; k < kp; k += 64) {
            float v = 0.f;
            if (valid && k < d) {
                v = x[r * (int64_t)d + k] * scale;
                nb2 += v * v;
            }
            if (!aug || k < d || k >= d + 2) img[(((k >> 4) * 2 + ((k >> 3) & 1)) * 32 + pr) * 8 + (k & 7)] = (_Float16)v;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nb2 += __shfl_xor(nb2, off);
        if (aug && lane < 2) {
            const float h = 0.5f * nb2;
            const _Float16 hi = (_Float16)h;
            const _Float16 lo = (_Float16)(h - (float)hi);
            _Float16 e = (_Float16)0.f;
            if (valid) e = (role == 0) ? (_Float16)1.0f : (lane == 0 ? (_Float16)(-(float)hi) : (_Float16)(-(float)lo));
            const int k = d + lane;
            img[(((k >> 4) * 2 + ((k >> 3) & 1)) * 32 + pr) * 8 + (k & 7)] = e;
        }
        // query role: "not a zero row" for match_select_kernel; map role without columns: the row bias
        if (lane == 0) inv_out[r] = (!aug && role == 1) ? (valid ? -0.5f * nb2 : 0.0f) : 1.0f;
    }
    __syncthreads();
    const int units = (kp >> 4) * 64;
    uint4* dst = tiles + (int64_t)tile * units;
    const uint4* src = reinterpret_cast<const uint4*>(smem);
    for (int u = threadIdx.x; u < units; u += 256) {
        const uint4 t = src[u];
        unsigned* o = reinterpret_cast<unsigned*>(dst + u);
        __builtin_nontemporal_store(t.x, o);
        __builtin_nontemporal_store(t.y, o + 1);
        __builtin_nontemporal_store(t.z, o + 2);
        __builtin_nontemporal_store(t.w, o + 3);
    }
}

// oracle order: acc += (double(a_k) - double(b_k))^2, k ascending
__device__ __forceinline__ double l2_dist_f64(const float* __restrict__ qa, const float* __restrict__ brow, int d) {
    double acc = 0.0;
    for (int k = 0; k < d; ++k) {
        const double t = (double)qa[k] - (double)brow[k];
        acc = acc + t * t;
    }
    return acc;
}

// exact decision among the candidates of match_select_kernel: one wave per query
__global__ __launch_bounds__(256) void l2_rescore_kernel(const float* __restrict__ q, const float* __restrict__ b, int64_t n,
                                                         int64_t m, int d, const int* __restrict__ cand_cnt,
                                                         const unsigned* __restrict__ cand, int cap, int64_t* __restrict__ nn_out,
                                                         double* __restrict__ d2_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    float* qa = reinterpret_cast<float*>(smem) + wave * 2 * d;
    float* bb = qa + d;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
    if (qi >= n) return;
    const int cnt = cand_cnt[qi];
    if (cnt < 0) return;  // handled by nn_l2_kernel (candidate overflow)
    for (int k = lane; k < d; k += 64) qa[k] = q[qi * (int64_t)d + k];
    __builtin_amdgcn_wave_barrier();
    double best = 0.0;  // negated distance: wave_argmax picks the smallest distance, ties -> lowest index
    long long bj = -1;
    for (int e = 0; e < cnt; ++e) {
        const unsigned ce = cand[(size_t)qi * cap + e];
        const long long base = (long long)(ce >> 8) * CHUNK_ROWS;
        if (ce & 128u) {
            for (int li = lane; li < CHUNK_ROWS; li += 64) {
                const long long j = base + li;
                if (j < m) {
                    const double s = -l2_dist_f64(qa, b + j * (int64_t)d, d);
                    if (bj < 0 || s > best || (s == best && j < bj)) {
                        best = s;
                        bj = j;
                    }
                }
            }
        } else {
            const long long j = base + (ce & 127u);
            if (j < m) {
                __builtin_amdgcn_wave_barrier();
                for (int k = lane; k < d; k += 64) bb[k] = b[j * (int64_t)d + k];
                __builtin_amdgcn_wave_barrier();
                const double s = -l2_dist_f64(qa, bb, d);  // same chain in every lane (LDS broadcast reads)
                if (bj < 0 || s > best || (s == best && j < bj)) {
                    best = s;
                    bj = j;
                }
            }
        }
    }
    wave_argmax(best, bj);
    if (lane == 0) {
        nn_out[qi] = bj;
        if (d2_out) d2_out[qi] = -best;
    }
}


}  // namespace

// d <= 510: two appended columns; 510 < d <= 768: row bias, no extra columns; wider: 0 (all-pairs fp64)
inline bool l2_aug(int d) { return d + 2 <= 512; }
// (the row-bias form exists in match_coarse_r_kernel only, i.e. for K = 640 and 768: d = 511, 512 pad to 640)
inline int l2_padded_k(int d) { return l2_aug(d) ? (d + 2 + 127) / 128 * 128 : (d <= 640 ? 640 : (d <= 768 ? 768 : 0)); }

struct L2Ws {
    unsigned* max_bits;
    void* prep[4];  // a as query, b as map, b as query, a as map
    void* search[2];
    size_t bytes;
};
inline L2Ws carve_l2(void* p, int64_t n, int64_t m, int d, bool mutual) {
    VfmCarver c(p);
    L2Ws w;
    const int kp = l2_padded_k(d);
    w.max_bits = c.take<unsigned>(64);
    const size_t pa = vfm_match_prepared_bytes(n, kp), pb = vfm_match_prepared_bytes(m, kp);
    w.prep[0] = c.take<unsigned char>(pa);
    w.prep[1] = c.take<unsigned char>(pb);
    w.search[0] = c.take<unsigned char>(carve_search(nullptr, n, m).bytes);
    w.prep[2] = w.prep[3] = w.search[1] = nullptr;
    if (mutual) {
        w.prep[2] = c.take<unsigned char>(pb);
        w.prep[3] = c.take<unsigned char>(pa);
        w.search[1] = c.take<unsigned char>(carve_search(nullptr, m, n).bytes);
    }
    w.bytes = c.used();
    return w;
}

int l2_prepare(const float* x, int64_t rows, int d, int kp, const unsigned* max_bits, int role, void* prepared, hipStream_t st) {
    Prepared p = carve_prepared(prepared, rows, kp);
    hipLaunchKernelGGL(l2_prep_kernel, dim3((unsigned)(rows_padded(rows) / TILE_ROWS)), dim3(256), (size_t)kp * 64, st, x, rows, d, kp,
                       max_bits, role, l2_aug(d) ? 1 : 0, p.inv, p.tiles);
    VFM_CHECK_LAUNCH("l2_prep_kernel");
    return VFM_OK;
}

// one direction: every row of q (n x d) among b (m x d)
int l2_search(const float* q, void* qprep, int64_t n, const float* b, void* bprep, int64_t m, int d, int kp, int64_t* nn,
              double* d2, void* ws, hipStream_t st) {
    if (int rc = do_search_coarse(qprep, n, bprep, m, kp, ws, st, !l2_aug(d))) return rc;
    Prepared Q = carve_prepared(qprep, n, kp);
    SearchWs w = carve_search(ws, n, m);
    const CoarseArgs a = coarse_args(Q, carve_prepared(bprep, m, kp), w, n, m, coarse_qblock(kp));
    if (int rc = launch_select_dense(w, a, Q.inv, n, st)) return rc;
    hipLaunchKernelGGL(l2_rescore_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)d * 4 * 8, st, q, b, n, m, d, w.cand_cnt,
                       w.cand, w.cap, nn, d2);
    VFM_CHECK_LAUNCH("l2_rescore_kernel");
    const size_t lds = (((size_t)d * 4 + 15) & ~(size_t)15) + 64;
    hipLaunchKernelGGL(nn_l2_kernel, dim3(256), dim3(256), lds, st, q, n, b, m, d, w.fb_list, w.fb_count, nn, d2);
    VFM_CHECK_LAUNCH("nn_l2_kernel(fallback)");
    return VFM_OK;
}

}  // namespace vfmm

using namespace vfmm;

VFM_EXPORT size_t vfm_match_mutual_l2_workspace_bytes(int64_t n, int64_t m, int d, int prec_mode, int mutual) {
    if (prec_mode == VFM_MATCH_EXACT || l2_padded_k(d) == 0 || n <= 0 || m <= 0) return 256;
    return carve_l2(nullptr, n, m, d, mutual != 0).bytes;
}

VFM_EXPORT int vfm_match_mutual_l2(const float* a, int64_t n, const float* b, int64_t m, int d, int prec_mode,
                                   int64_t* nn_ab, double* d2_ab, int64_t* nn_ba, void* ws, size_t ws_bytes,
                                   vfm_stream_t stream) {
    VFM_CHECK_ARG(n > 0 && m > 0 && d > 0 && a && b && nn_ab, "mutual_l2: bad arguments");
    VFM_CHECK_ARG(prec_mode == VFM_MATCH_FAST || prec_mode == VFM_MATCH_EXACT, "mutual_l2: unknown prec_mode %d", prec_mode);
    VFM_CHECK_ARG(m < (1ll << 31) - 256 && n < (1ll << 31) - 256, "mutual_l2: more than 2^31 rows");
    hipStream_t st = (hipStream_t)stream;
    const int kp = l2_padded_k(d);
    if (prec_mode == VFM_MATCH_EXACT || kp == 0) {
        // all-pairs fp64 (also the path for descriptors wider than 768)
        const size_t lds = (((size_t)d * 4 + 15) & ~(size_t)15) + 64;
        hipLaunchKernelGGL(nn_l2_kernel, dim3((unsigned)(n < 8192 ? n : 8192)), dim3(256), lds, st, a, n, b, m, d,
                           (const int*)nullptr, (const int*)nullptr, nn_ab, d2_ab);
        if (nn_ba)
            hipLaunchKernelGGL(nn_l2_kernel, dim3((unsigned)(m < 8192 ? m : 8192)), dim3(256), lds, st, b, m, a, n, d,
                               (const int*)nullptr, (const int*)nullptr, nn_ba, (double*)nullptr);
        VFM_CHECK_LAUNCH("nn_l2_kernel");
        return VFM_OK;
    }
    VFM_CHECK_ARG(ws, "mutual_l2: workspace required in FAST mode");
    if (ws_bytes < vfm_match_mutual_l2_workspace_bytes(n, m, d, prec_mode, nn_ba != nullptr))
        return vfm_fail(VFM_EWORKSPACE, "mutual_l2: workspace too small");
    L2Ws w = carve_l2(ws, n, m, d, nn_ba != nullptr);
    VFM_CHECK_HIP(hipMemsetAsync(w.max_bits, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(l2_maxnorm_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, a, n, d, w.max_bits);
    hipLaunchKernelGGL(l2_maxnorm_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, b, m, d, w.max_bits);
    VFM_CHECK_LAUNCH("l2_maxnorm_kernel");
    if (int rc = l2_prepare(a, n, d, kp, w.max_bits, 0, w.prep[0], st)) return rc;
    if (int rc = l2_prepare(b, m, d, kp, w.max_bits, 1, w.prep[1], st)) return rc;
    if (int rc = l2_search(a, w.prep[0], n, b, w.prep[1], m, d, kp, nn_ab, d2_ab, w.search[0], st)) return rc;
    if (nn_ba) {
        if (int rc = l2_prepare(b, m, d, kp, w.max_bits, 0, w.prep[2], st)) return rc;
        if (int rc = l2_prepare(a, n, d, kp, w.max_bits, 1, w.prep[3], st)) return rc;
        if (int rc = l2_search(b, w.prep[2], m, a, w.prep[3], n, d, kp, nn_ba, nullptr, w.search[1], st)) return rc;
    }
    return VFM_OK;
}

