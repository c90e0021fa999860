// ransac.hip -- correspondence RANSAC + Kabsch on MI355X (gfx950).
//
// Replaces open3d.pipelines.registration.registration_ransac_based_on_correspondence as called
// at registration_node.py:319-327 (Open3D 0.18: draw 3 correspondences with replacement,
// Eigen::umeyama without scaling, score |T s - t|^2 < d^2 over ALL correspondences, keep the
// hypothesis with higher fitness, then lower RMSE; confidence 1 => every iteration runs) and
// Eigen::umeyama / pointdsc.common.rigid_transform_3d (pointdsc/common.py:7-47).
//
// Structure: cheap, PROVEN bounds on every hypothesis' inlier count and RMSE
//   ransac_center_kernel + ransac_moment_kernel   closed form from the stream's second moments when every
//                                                 correspondence is provably an inlier (the reference's
//                                                 max_correspondence_distance = 10000 m)
//   ransac_coarse_kernel                          point-wise fp32 otherwise
// -> ransac_select_{rmin,list}_kernel: the hypotheses the bounds cannot rule out
// -> ransac_exact_list_kernel: those are re-scored in fp64 in the oracle's operation and summation order
//    (ransac_score_kernel does that for ALL hypotheses: reference semantics, A/B switch, overflow fallback)
// -> ransac_final_kernel: total order (fitness desc, rmse asc, id asc), pose of the winner; ransac_mask_kernel.
// Exact arithmetic is fp64, compiled with -ffp-contract=off, written as the exact operation sequence of
// oracle/vfm_oracle.c so that poses, masks and the winning hypothesis are bit-identical to the oracle's.
#include <atomic>

#include "common.h"

namespace {

constexpr int JACOBI_SWEEPS = 6;

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n1 = lo1;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        const uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// rotation from the 3x3 cross-covariance: one-sided Jacobi, fixed sweeps, only + - * / sqrt.
__device__ bool rot_from_sigma(const double S[9], double R[9]) {
    double G[3][3], V[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            G[r][c] = S[r * 3 + c];
            V[r][c] = (r == c) ? 1.0 : 0.0;
        }
#pragma unroll 1
    for (int sweep = 0; sweep < JACOBI_SWEEPS; ++sweep) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int p = (e == 2) ? 1 : 0;
            const int q = (e == 0) ? 1 : 2;
            const double alpha = (G[0][p] * G[0][p] + G[1][p] * G[1][p]) + G[2][p] * G[2][p];
            const double beta = (G[0][q] * G[0][q] + G[1][q] * G[1][q]) + G[2][q] * G[2][q];
            const double gamma = (G[0][p] * G[0][q] + G[1][p] * G[1][q]) + G[2][p] * G[2][q];
            if (gamma == 0.0) continue;
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double az = fabs(zeta);
            double tt = 1.0 / (az + sqrt(1.0 + zeta * zeta));
            if (zeta < 0.0) tt = -tt;
            const double c = 1.0 / sqrt(1.0 + tt * tt);
            const double s = c * tt;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double gp = G[r][p], gq = G[r][q];
                G[r][p] = c * gp - s * gq;
                G[r][q] = s * gp + c * gq;
                const double vp = V[r][p], vq = V[r][q];
                V[r][p] = c * vp - s * vq;
                V[r][q] = s * vp + c * vq;
            }
        }
    }
    double nn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) nn[c] = (G[0][c] * G[0][c] + G[1][c] * G[1][c]) + G[2][c] * G[2][c];
    // two dominant columns, ties -> lowest column index (static indexing only: select by value)
    int i1 = 0;
    if (nn[1] > nn[i1 == 0 ? 0 : 1]) i1 = 1;
    {
        const double cur = (i1 == 0) ? nn[0] : nn[1];
        if (nn[2] > cur) i1 = 2;
    }
    int i2;
    {
        // candidates are the two columns != i1, scanned in ascending order, strict '>' to replace
        const int ca = (i1 == 0) ? 1 : 0;
        const int cb = (i1 == 2) ? 1 : 2;
        const double na = (ca == 0) ? nn[0] : nn[1];
        const double nb = (cb == 1) ? nn[1] : nn[2];
        i2 = (nb > na) ? cb : ca;
    }
    const double n1 = (i1 == 0) ? nn[0] : ((i1 == 1) ? nn[1] : nn[2]);
    const double n2 = (i2 == 0) ? nn[0] : ((i2 == 1) ? nn[1] : nn[2]);
    if (!(n1 > 0.0)) return false;
    if (!(n2 > n1 * 1e-20)) return false;
    const double s1 = sqrt(n1), s2 = sqrt(n2);
    double u1[3], u2[3], u3[3], v1[3], v2[3], v3[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double g1 = (i1 == 0) ? G[r][0] : ((i1 == 1) ? G[r][1] : G[r][2]);
        const double g2 = (i2 == 0) ? G[r][0] : ((i2 == 1) ? G[r][1] : G[r][2]);
        u1[r] = g1 / s1;
        u2[r] = g2 / s2;
        v1[r] = (i1 == 0) ? V[r][0] : ((i1 == 1) ? V[r][1] : V[r][2]);
        v2[r] = (i2 == 0) ? V[r][0] : ((i2 == 1) ? V[r][1] : V[r][2]);
    }
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
    u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
    u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    v3[0] = v1[1] * v2[2] - v1[2] * v2[1];
    v3[1] = v1[2] * v2[0] - v1[0] * v2[2];
    v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = (u1[r] * v1[c] + u2[r] * v2[c]) + u3[r] * v3[c];
    return true;
}

// Kabsch on n weighted pairs; A, B: n x 3 with the given element stride. T: 12 doubles
// (rows of [R|t]).  Same operation order as orc_kabsch.
__device__ bool kabsch_rows(const double* A, const double* B, const double* w, int64_t n, double denom_eps,
                            double T[12]) {
    double ma[3] = {0.0, 0.0, 0.0}, mb[3] = {0.0, 0.0, 0.0}, sw = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double wi = w ? w[i] : 1.0;
        sw = sw + wi;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ma[c] = ma[c] + wi * A[i * 3 + c];
            mb[c] = mb[c] + wi * B[i * 3 + c];
        }
    }
    const double inv = 1.0 / (sw + denom_eps);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ma[c] = ma[c] * inv;
        mb[c] = mb[c] * inv;
    }
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        const double wi = w ? w[i] : 1.0;
        double ad[3], bd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ad[c] = A[i * 3 + c] - ma[c];
            bd[c] = B[i * 3 + c] - mb[c];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) S[r * 3 + c] = S[r * 3 + c] + (wi * bd[r]) * ad[c];
    }
    const double invs = 1.0 / sw;
#pragma unroll
    for (int k = 0; k < 9; ++k) S[k] = S[k] * invs;
    double R[9];
    if (!rot_from_sigma(S, R)) return false;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        T[r * 4 + 0] = R[r * 3 + 0];
        T[r * 4 + 1] = R[r * 3 + 1];
        T[r * 4 + 2] = R[r * 3 + 2];
        T[r * 4 + 3] = mb[r] - ((R[r * 3 + 0] * ma[0] + R[r * 3 + 1] * ma[1]) + R[r * 3 + 2] * ma[2]);
    }
    return true;
}

__device__ __forceinline__ double err2(const double T[12], double sx, double sy, double sz, double tx, double ty,
                                       double tz) {
    const double x = ((T[0] * sx + T[1] * sy) + T[2] * sz) + T[3];
    const double y = ((T[4] * sx + T[5] * sy) + T[6] * sz) + T[7];
    const double z = ((T[8] * sx + T[9] * sy) + T[10] * sz) + T[11];
    const double dx = x - tx, dy = y - ty, dz = z - tz;
    return (dx * dx + dy * dy) + dz * dz;
}

// pts[i] = {src[corres[i][0]], tgt[corres[i][1]]}: the wave-uniform stream of the scoring loop
__global__ __launch_bounds__(256) void ransac_gather_kernel(const double* __restrict__ src, const double* __restrict__ tgt,
                                                            const int32_t* __restrict__ corres,
                                                            const int64_t* __restrict__ count_dev, int64_t c_max,
                                                            double* __restrict__ pts, int64_t ns, int64_t nt,
                                                            int32_t* __restrict__ bad_out) {
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= C) return;
    int64_t a = corres[2 * i], b = corres[2 * i + 1];
    if (bad_out && (a < 0 || a >= ns || b < 0 || b >= nt)) {   // vfm_ransac_corr_bounded: flagged, read as row 0
        *bad_out = 1;
        a = b = 0;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pts[6 * i + c] = src[3 * a + c];
        pts[6 * i + 3 + c] = tgt[3 * b + c];
    }
}

__device__ __forceinline__ bool sample_T(const double* __restrict__ pts, int64_t C, uint32_t hyp, uint64_t seed,
                                         double T[12]) {
    uint32_t r[4];
    philox4x32_10(hyp, 0u, 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    double A[9], B[9];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint64_t pick = ((uint64_t)r[j] * (uint64_t)C) >> 32;
        const double* p = pts + 6 * pick;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            A[j * 3 + c] = p[c];
            B[j * 3 + c] = p[3 + c];
        }
    }
    return kabsch_rows(A, B, nullptr, 3, 0.0, T);
}

struct HypScore {
    double fit;
    double rmse;
    int32_t hyp;
};

// a strictly better than b under Open3D's IsBetterRANSACThan + earliest-hypothesis tie-break;
// entries with hyp < 0 are empty.
__device__ __forceinline__ bool better(double af, double ar, int ah, double bf, double br, int bh) {
    if (ah < 0) return false;
    if (bh < 0) return true;
    if (af > bf) return true;
    if (af < bf) return false;
    if (ar < br) return true;
    if (ar > br) return false;
    return ah < bh;
}

constexpr int SCORE_CHUNK = 128;  // correspondences staged in LDS per step (6 KiB)

// Exact (fp64, oracle operation order) scoring of ALL n_iter hypotheses, one lane per hypothesis: the
// semantics every other path must reproduce (vfm_debug_set_ransac_exact_only), and the fallback when the
// candidate list overflowed (gate != NULL: runs only if *gate != 0).
__global__ __launch_bounds__(64) void ransac_score_kernel(const double* __restrict__ pts,
                                                          const int64_t* __restrict__ count_dev, int64_t c_max,
                                                          double max_d2, int32_t n_iter, uint64_t seed,
                                                          const int32_t* __restrict__ gate,
                                                          HypScore* __restrict__ block_best) {
    // one wavefront per workgroup: its private LDS double buffer holds the correspondence stream
    __shared__ __attribute__((aligned(16))) double lbuf[2][SCORE_CHUNK * 6];
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    const int lane = threadIdx.x;
    const int32_t slot = (int32_t)(blockIdx.x * 64 + lane);
    const int32_t limit = n_iter;
    const bool enabled = !gate || *gate != 0;  // gate: run only if the flag is set (candidate-list overflow)
    const int32_t h = (enabled && slot < limit) ? slot : n_iter;
    double fit = 0.0, rmse = 0.0;
    int hyp = -1;
    double T[12];
    const bool wave_has_work = enabled && (C >= 3) && ((int32_t)(blockIdx.x * 64) < limit);
    bool live = false;
    if (C >= 3 && h < n_iter) live = sample_T(pts, C, (uint32_t)h, seed, T);
    if (!live) {
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = 0.0;
    }
    if (wave_has_work) {
        int64_t good = 0;
        double e2 = 0.0;
        // chunk c covers correspondences [c*128, c*128+128): 768 doubles = 64 lanes x 6 double2
        const int64_t nchunks = (C + SCORE_CHUNK - 1) / SCORE_CHUNK;
        double2 pre[6];
        auto fetch = [&](int64_t c) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int64_t e = c * (SCORE_CHUNK * 6) + (int64_t)(j * 64 + lane) * 2;  // element index
                pre[j] = (e + 1 < C * 6) ? *reinterpret_cast<const double2*>(pts + e) : make_double2(0.0, 0.0);
            }
        };
        auto stash = [&](int b) {
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<double2*>(&lbuf[b][(j * 64 + lane) * 2]) = pre[j];
        };
        fetch(0);
        stash(0);
        for (int64_t c = 0; c < nchunks; ++c) {
            const int b = (int)(c & 1);
            if (c + 1 < nchunks) fetch(c + 1);  // global loads in flight during the scoring below
            __builtin_amdgcn_wave_barrier();
            const int64_t base = c * SCORE_CHUNK;
            const int cnt = (int)min((int64_t)SCORE_CHUNK, C - base);
            const double* lp = lbuf[b];
            // 4 correspondences per trip: independent fp64 chains (the kernel is bound by the
            // dependent-issue latency of the fp64 ALU at one wave per SIMD); (good, e2) are folded
            // in the original order, so the accumulation order of the oracle / Open3D is kept.
            int i = 0;
            for (; i + 4 <= cnt; i += 4) {
                double d2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double* p = lp + 6 * (i + u);  // same address in every lane: LDS broadcast
                    d2[u] = err2(T, p[0], p[1], p[2], p[3], p[4], p[5]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (d2[u] < max_d2) {
                        good++;
                        e2 = e2 + d2[u];
                    }
            }
            for (; i < cnt; ++i) {
                const double* p = lp + 6 * i;
                const double d2 = err2(T, p[0], p[1], p[2], p[3], p[4], p[5]);
                if (d2 < max_d2) {
                    good++;
                    e2 = e2 + d2;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (c + 1 < nchunks) stash(b ^ 1);
        }
        if (live && good > 0) {  // fitness 0 can never beat the initial (0, 0) result
            fit = (double)good / (double)C;
            rmse = sqrt(e2 / (double)good);
            hyp = h;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double of = __shfl_xor(fit, off), orr = __shfl_xor(rmse, off);
        const int oh = __shfl_xor(hyp, off);
        if (better(of, orr, oh, fit, rmse, hyp)) {
            fit = of;
            rmse = orr;
            hyp = oh;
        }
    }
    if (threadIdx.x == 0) {
        HypScore s;
        s.fit = fit;
        s.rmse = rmse;
        s.hyp = hyp;
        block_best[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void ransac_final_kernel(const double* __restrict__ pts,
                                                           const int64_t* __restrict__ count_dev, int64_t c_max,
                                                           uint64_t seed, const HypScore* __restrict__ block_best,
                                                           int nblocks, double* __restrict__ T_out,
                                                           double* __restrict__ fitness_out, double* __restrict__ rmse_out,
                                                           int32_t* __restrict__ best_hyp_out) {
    __shared__ double sf[4], sr[4];
    __shared__ int sh[4];
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    double fit = 0.0, rmse = 0.0;
    int hyp = -1;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        const HypScore s = block_best[b];
        if (better(s.fit, s.rmse, s.hyp, fit, rmse, hyp)) {
            fit = s.fit;
            rmse = s.rmse;
            hyp = s.hyp;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double of = __shfl_xor(fit, off), orr = __shfl_xor(rmse, off);
        const int oh = __shfl_xor(hyp, off);
        if (better(of, orr, oh, fit, rmse, hyp)) {
            fit = of;
            rmse = orr;
            hyp = oh;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        sf[threadIdx.x >> 6] = fit;
        sr[threadIdx.x >> 6] = rmse;
        sh[threadIdx.x >> 6] = hyp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (better(sf[w], sr[w], sh[w], sf[0], sr[0], sh[0])) {
                sf[0] = sf[w];
                sr[0] = sr[w];
                sh[0] = sh[w];
            }
        double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        if (sh[0] >= 0) sample_T(pts, C, (uint32_t)sh[0], seed, T);
        for (int k = 0; k < 12; ++k) T_out[k] = T[k];
        T_out[12] = 0.0;
        T_out[13] = 0.0;
        T_out[14] = 0.0;
        T_out[15] = 1.0;
        *fitness_out = (sh[0] >= 0) ? sf[0] : 0.0;
        *rmse_out = (sh[0] >= 0) ? sr[0] : 0.0;
        *best_hyp_out = sh[0];
    }
}

__global__ __launch_bounds__(256) void ransac_mask_kernel(const double* __restrict__ pts,
                                                          const int64_t* __restrict__ count_dev, int64_t c_max,
                                                          double max_d2, const double* __restrict__ T_in,
                                                          const int32_t* __restrict__ best_hyp,
                                                          uint8_t* __restrict__ mask) {
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= c_max) return;
    uint8_t m = 0;
    if (i < C && *best_hyp >= 0) {
        double T[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = T_in[k];
        const double* p = pts + 6 * i;
        m = (err2(T, p[0], p[1], p[2], p[3], p[4], p[5]) < max_d2) ? 1 : 0;
    }
    mask[i] = m;
}

__global__ __launch_bounds__(64) void kabsch_batched_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                            const double* __restrict__ w, int64_t b, int64_t n,
                                                            double denom_eps, double* __restrict__ T_out,
                                                            int32_t* __restrict__ valid) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= b) return;
    double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const bool ok = kabsch_rows(A + i * n * 3, B + i * n * 3, w ? w + i * n : nullptr, n, denom_eps, T);
    if (!ok) {
        const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        for (int k = 0; k < 12; ++k) T[k] = I[k];
    }
    for (int k = 0; k < 12; ++k) T_out[i * 16 + k] = T[k];
    T_out[i * 16 + 12] = 0.0;
    T_out[i * 16 + 13] = 0.0;
    T_out[i * 16 + 14] = 0.0;
    T_out[i * 16 + 15] = 1.0;
    if (valid) valid[i] = ok ? 1 : 0;
}


// ---------------------------------------------------------------------------------------------
// fp32 coarse scoring with a proven error bound (the same idea as the matcher's fp16 coarse pass):
// every hypothesis is scored in fp32 FMA arithmetic (~6x cheaper than the fp64 no-FMA sequence),
// together with bounds [n_lo, n_hi] on its inlier count and [r_lo, r_hi] on its inlier RMSE;
// only hypotheses that these bounds cannot rule out are re-scored exactly by ransac_score_kernel.
// The winner (and therefore T, fitness, rmse, mask) is the one the all-fp64 pass would return.
//
// Bound.  With centred clouds s' = s - cs, q' = q - cq, t' = t + R cs - cq (exact identity
// R s + t - q = R s' + t' - q'), M = max |s'|, |q'| and u = 2^-24: every component of the fp32
// residual d~ = fl(R~ s~' + t~' - q~') differs from the exact one by at most
//     delta = u (19 M + 5 |t'|_inf)  <=  u (32 M + 8 |t'|_inf)      (input roundings + 4 fp32 ops),
// hence | |d~| - |d| | <= eta = sqrt(3) delta.  A correspondence with d~^2 < (max_dist - eta)^2 is
// certainly an inlier, one with d~^2 >= (max_dist + eta)^2 certainly is not; over a certain set
// RMSE(d) lies within eta of RMSE(d~) (Minkowski); uncertain members (all at distance ~max_dist)
// widen the lower bound by another 2 eta.  fp32 chunk sums (128 terms) are covered by 1e-4 slack.
// ---------------------------------------------------------------------------------------------
constexpr int COARSE_CHUNK = 128;
constexpr int COARSE_WAVES = 8;
constexpr int CAND_MAX = 2048;

struct RansacStats {
    double cs[3], cq[3];  // centroids of the gathered source / target points
    double M;             // max |centred coordinate|
    // second moments of the centred stream sigma_i = s_i - cs, kappa_i = q_i - cq (closed-form score
    // of the all-inlier regime, ransac_moment_kernel)
    double Sss[6];         // sum sigma_b sigma_c: xx, xy, xz, yy, yz, zz
    double K[9];           // sum kappa_a sigma_b, row-major [a][b]
    double Akk;            // sum |kappa|^2
    double sbar[3], kbar[3];  // sum sigma, sum kappa (~0)
    double Xs, Xq;         // max |raw coordinate| of source / target
    double gm;             // terms per partial sum of the moment reductions (rounding-error constant)
};

// one workgroup: centroids, extent, and the centred fp32 copy of the correspondence stream
__device__ __forceinline__ void ransac_center_body(const double* __restrict__ pts, int64_t C, RansacStats* __restrict__ stats,
                                                   float* __restrict__ pts32, double (&red)[6][1024], double (&cen)[6]) {
    const int t = threadIdx.x;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = t; i < C; i += 1024)
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] += pts[6 * i + c];
#pragma unroll
    for (int c = 0; c < 6; ++c) red[c][t] = acc[c];
    __syncthreads();
    for (int stride = 512; stride >= 1; stride >>= 1) {
        if (t < stride)
#pragma unroll
            for (int c = 0; c < 6; ++c) red[c][t] += red[c][t + stride];
        __syncthreads();
    }
    if (t < 6) cen[t] = (C > 0) ? red[t][0] / (double)C : 0.0;
    __syncthreads();
    double m = 0.0, xs = 0.0, xq = 0.0;
    double mom[22];  // Sss (6), K (9), Akk, sbar (3), kbar (3)
#pragma unroll
    for (int k = 0; k < 22; ++k) mom[k] = 0.0;
    // (measured and dropped in round 6: the loads of four strides issued together -- 48 more registers -- made the kernel slower,
    // 0.193 against 0.185 ms for the stage: tools/time_ransac.py)
    for (int64_t i = t; i < C; i += 1024) {
        double v[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double raw = pts[6 * i + c];
            v[c] = raw - cen[c];
            pts32[6 * i + c] = (float)v[c];
            m = fmax(m, fabs(v[c]));
            if (c < 3) xs = fmax(xs, fabs(raw)); else xq = fmax(xq, fabs(raw));
        }
        mom[0] += v[0] * v[0]; mom[1] += v[0] * v[1]; mom[2] += v[0] * v[2];
        mom[3] += v[1] * v[1]; mom[4] += v[1] * v[2]; mom[5] += v[2] * v[2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) mom[6 + 3 * a + b] += v[3 + a] * v[b];
        mom[15] += (v[3] * v[3] + v[4] * v[4]) + v[5] * v[5];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            mom[16 + c] += v[c];
            mom[19 + c] += v[3 + c];
        }
    }
    // wave butterflies (order-symmetric, deterministic), then 16 partials per quantity in LDS
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 22; ++k) mom[k] += __shfl_xor(mom[k], off);
        m = fmax(m, __shfl_xor(m, off));
        xs = fmax(xs, __shfl_xor(xs, off));
        xq = fmax(xq, __shfl_xor(xq, off));
    }
    __syncthreads();  // red[] is free again
    if ((t & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 22; ++k) red[0][(t >> 6) * 32 + k] = mom[k];
        red[0][(t >> 6) * 32 + 22] = m;
        red[0][(t >> 6) * 32 + 23] = xs;
        red[0][(t >> 6) * 32 + 24] = xq;
    }
    __syncthreads();
    if (t < 25) {
        double acc1 = red[0][t];
        for (int w = 1; w < 16; ++w) acc1 = (t < 22) ? acc1 + red[0][w * 32 + t] : fmax(acc1, red[0][w * 32 + t]);
        red[1][t] = acc1;
    }
    __syncthreads();
    if (t == 0) {
        for (int c = 0; c < 3; ++c) {
            stats->cs[c] = cen[c];
            stats->cq[c] = cen[3 + c];
        }
        stats->M = red[1][22];
        for (int k = 0; k < 6; ++k) stats->Sss[k] = red[1][k];
        for (int k = 0; k < 9; ++k) stats->K[k] = red[1][6 + k];
        stats->Akk = red[1][15];
        for (int k = 0; k < 3; ++k) {
            stats->sbar[k] = red[1][16 + k];
            stats->kbar[k] = red[1][19 + k];
        }
        stats->Xs = red[1][23];
        stats->Xq = red[1][24];
        stats->gm = (double)((C + 1023) / 1024) + 32.0;  // per-thread terms + butterfly (6) + partials (16) + products / centring
    }
}
struct SelectState;
__device__ __forceinline__ void ransac_sel_reset(SelectState* sel);
__global__ __launch_bounds__(1024) void ransac_center_kernel(const double* __restrict__ pts,
                                                             const int64_t* __restrict__ count_dev, int64_t c_max,
                                                             RansacStats* __restrict__ stats, float* __restrict__ pts32,
                                                             SelectState* __restrict__ sel_to_reset) {
    __shared__ double red[6][1024];
    __shared__ double cen[6];
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    if (sel_to_reset && threadIdx.x == 0) ransac_sel_reset(sel_to_reset);   // (chain 2: ransac_sel_init_kernel's launch)
    ransac_center_body(pts, C, stats, pts32, red, cen);
}

struct SelectState {           // zeroed per call (Rbits = +inf)
    int F;                     // max certain inlier count
    int count;                 // candidates appended
    unsigned long long Rbits;  // min r_hi among hypotheses whose count is certainly F (bits of a double >= 0)
    int overflow;              // candidate list overflowed -> score everything exactly
    int unsure;                // some hypothesis is not certainly all-inlier -> point-wise coarse pass needed
    int done;                  // fused chain: workgroups of the point-wise pass that have finished (the last one recomputes Rbits)
};

__device__ __forceinline__ void ransac_sel_reset(SelectState* sel) {
    sel->F = 0;
    sel->count = 0;
    sel->Rbits = 0x7FF0000000000000ull;  // +inf
    sel->overflow = 0;
    sel->unsure = 0;
    sel->done = 0;
}
__global__ void ransac_sel_init_kernel(SelectState* sel) { ransac_sel_reset(sel); }

// Fused chain (round 6; VERDICT r5 item 3a: the stage is a chain of dependent launches, 11 of them, 0.18 ms for 0.02 ms of arithmetic):
// gather + select-state + centring / moments in ONE workgroup (the stream is 10^4 correspondences: 480 KB).
__global__ __launch_bounds__(1024) void ransac_prepare_kernel(const double* __restrict__ src, const double* __restrict__ tgt,
                                                              const int32_t* __restrict__ corres, const int64_t* __restrict__ count_dev,
                                                              int64_t c_max, double* __restrict__ pts, int64_t ns, int64_t nt,
                                                              int32_t* __restrict__ bad_out, RansacStats* __restrict__ stats,
                                                              float* __restrict__ pts32, SelectState* __restrict__ sel) {
    __shared__ double red[6][1024];
    __shared__ double cen[6];
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    if (threadIdx.x == 0) {
        sel->F = 0;
        sel->count = 0;
        sel->Rbits = 0x7FF0000000000000ull;
        sel->overflow = 0;
        sel->unsure = 0;
        sel->done = 0;
    }
    for (int64_t i = threadIdx.x; i < C; i += 1024) {
        int64_t a = corres[2 * i], b = corres[2 * i + 1];
        if (bad_out && (a < 0 || a >= ns || b < 0 || b >= nt)) {   // vfm_ransac_corr_bounded: flagged, read as row 0
            *bad_out = 1;
            a = b = 0;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pts[6 * i + c] = src[3 * a + c];
            pts[6 * i + 3 + c] = tgt[3 * b + c];
        }
    }
    __threadfence();     // the stream is read back below by other threads of this workgroup (and by the kernels behind it)
    __syncthreads();
    ransac_center_body(pts, C, stats, pts32, red, cen);
}

struct CoarseHyp {
    int32_t n_lo, n_hi;  // certain inliers / possible inliers (n_hi < 0: degenerate sample)
    double r_lo, r_hi;   // bounds on the inlier RMSE
};

// Closed-form coarse score for the regime the reference actually runs (registration_node.py:319-327
// passes max_correspondence_distance = 10000 m, so every correspondence is an inlier of every sane
// hypothesis): with sigma_i = s_i - cs, kappa_i = q_i - cq, tau = t + R cs - cq,
//   E(R,t) = sum_i |R sigma_i + tau - kappa_i|^2
//          = sum_a R_a^T Sss R_a + C |tau|^2 + Akk + 2 tau.(R sbar - kbar) - 2 sum_ab R_ab K_ab
// -- O(1) per hypothesis from the second moments of the stream instead of O(C).  It is only a
// PREFILTER: [E - eps, E + eps] provably contains the oracle's sequentially accumulated fp64 sum
// (derivation in DESIGN.md 4.3: moment/evaluation rounding <= u (gm + 80) B with B the sum of the
// magnitudes of all terms; the oracle's own residual rounding <= 48 u A sqrt(C E) with A the raw
// coordinate magnitude; its summation error <= (C + 8) u E), and the survivors are re-scored in the
// oracle's order by ransac_exact_list_kernel.  A hypothesis for which "every point is an inlier"
// cannot be proven (or max_dist <= 0) raises sel->unsure and the point-wise fp32 pass runs instead.
__global__ __launch_bounds__(64) void ransac_moment_kernel(const double* __restrict__ pts, const RansacStats* __restrict__ stats,
                                                           const int64_t* __restrict__ count_dev, int64_t c_max,
                                                           double max_d2, int32_t n_iter, uint64_t seed,
                                                           CoarseHyp* __restrict__ out, SelectState* __restrict__ sel, int rfuse) {
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    const int lane = threadIdx.x;
    const int32_t h = (int32_t)(blockIdx.x * 64 + lane);
    CoarseHyp o = CoarseHyp{0, -1, 0.0, 0.0};
    bool unsure = false, sure_live = false;
    if (C >= 3 && h < n_iter) {
        double T[12];
        if (sample_T(pts, C, (uint32_t)h, seed, T)) {
            const double u = 1.1102230246251565e-16;
            double tau[3], taumax = 0.0, tt = 0.0, tabs = 0.0, rmax = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                tau[r] = (T[4 * r + 3] + ((T[4 * r] * stats->cs[0] + T[4 * r + 1] * stats->cs[1]) + T[4 * r + 2] * stats->cs[2])) -
                         stats->cq[r];
                taumax = fmax(taumax, fabs(tau[r]));
                tt += tau[r] * tau[r];
                tabs = fmax(tabs, fabs(T[4 * r + 3]));
#pragma unroll
                for (int c = 0; c < 3; ++c) rmax = fmax(rmax, fabs(T[4 * r + c]));
            }
            // |residual component| <= 3 * 1.001 M + |tau| + M for every point
            const double reach = 4.01 * stats->M + taumax;
            const bool sure = (rmax <= 1.001) && (max_d2 > 0.0) && (3.0 * reach * reach * 1.001 < max_d2);
            if (!sure) {
                unsure = true;
            } else {
                const double* S = stats->Sss;
                double q1 = 0.0, rk = 0.0, t4 = 0.0;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const double x = T[4 * a], y = T[4 * a + 1], z = T[4 * a + 2];
                    q1 += (x * (S[0] * x + S[1] * y + S[2] * z) + y * (S[1] * x + S[3] * y + S[4] * z)) +
                          z * (S[2] * x + S[4] * y + S[5] * z);
                    rk += (x * stats->K[3 * a] + y * stats->K[3 * a + 1]) + z * stats->K[3 * a + 2];
                    t4 += tau[a] * (((x * stats->sbar[0] + y * stats->sbar[1]) + z * stats->sbar[2]) - stats->kbar[a]);
                }
                const double Cd = (double)C;
                const double E = (((q1 + Cd * tt) + stats->Akk) + 2.0 * t4) - 2.0 * rk;
                const double Ass = (S[0] + S[3]) + S[5];
                const double sbm = fmax(fmax(fabs(stats->sbar[0]), fabs(stats->sbar[1])), fabs(stats->sbar[2]));
                const double kbm = fmax(fmax(fabs(stats->kbar[0]), fabs(stats->kbar[1])), fabs(stats->kbar[2]));
                const double B = ((13.0 * Ass + 5.0 * stats->Akk) + 1.01 * Cd * tt) + 6.0 * taumax * (3.01 * sbm + kbm);
                const double eps1 = u * (stats->gm + 80.0) * B;
                const double Ep = fmax(E, 0.0) + eps1;
                const double A = (3.01 * stats->Xs + tabs) + stats->Xq;
                const double eps = (eps1 + (Cd + 8.0) * u * Ep) + 48.0 * u * A * sqrt(Cd * Ep);
                const double lo = fmax(0.0, E - eps), hi = fmax(0.0, E + eps);
                o.n_lo = (int32_t)C;
                o.n_hi = (int32_t)C;
                o.r_lo = sqrt(lo / Cd) * (1.0 - 1e-13);
                o.r_hi = sqrt(hi / Cd) * (1.0 + 1e-13);
                sure_live = true;
            }
        }
    }
    if (h < n_iter) out[h] = o;
    if (__any(unsure) && lane == 0) atomicExch(&sel->unsure, 1);
    if (__any(sure_live) && lane == 0) atomicMax(&sel->F, (int)C);
    if (rfuse) {
        // fused chain: R* = min r_hi over the hypotheses whose count is certainly F -- here F = C is known in advance (a hypothesis bounded
        // in closed form has every correspondence as an inlier, and no count exceeds C), so ransac_select_rmin_kernel's pass is this one.
        // (Should some workgroup raise `unsure`, the point-wise pass recomputes every bound and its last workgroup recomputes R*.)
        double r = sure_live ? o.r_hi : 1.7976931348623157e308;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) r = fmin(r, __shfl_xor(r, off));
        if (lane == 0 && r < 1.7976931348623157e308) atomicMin(&sel->Rbits, (unsigned long long)__double_as_longlong(r));
    }
}

// 512 threads = 8 waves per 64 hypotheses: wave w scores chunks w, w+8, ... of the stream (the coarse
// sums have no prescribed order), so 6 waves share a SIMD and hide each other's issue latency (a lone
// wave issues one instruction per ~4 cycles whatever its type).  Wave 0 derives the fp32 transform
// and the thresholds of each hypothesis once; partial results are combined in LDS in wave order.
__device__ __forceinline__ void ransac_coarse_block(const int hblock, const double* __restrict__ pts, const float* __restrict__ pts32,
                                                    const RansacStats* __restrict__ stats,
                                                    const int64_t* __restrict__ count_dev, int64_t c_max,
                                                    double max_dist, int32_t n_iter, uint64_t seed,
                                                    CoarseHyp* __restrict__ out, SelectState* __restrict__ sel) {
    __shared__ __attribute__((aligned(16))) float lbuf[COARSE_WAVES][2][COARSE_CHUNK * 6];
    __shared__ float hyp_f[14][64];  // R (9), t' (3), Lf, Hf per hypothesis
    __shared__ double hyp_eta[64];
    __shared__ int hyp_live[64];
    __shared__ int all_inliers_s;  // every correspondence is certainly an inlier of every hypothesis of this block
    __shared__ int part_n[COARSE_WAVES][2][64];
    __shared__ double part_e[COARSE_WAVES][2][64];
    if (sel->unsure == 0) return;  // ransac_moment_kernel bounded every hypothesis already (uniform)
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t h = (int32_t)(hblock * 64 + lane);
    if (C < 3) {
        if (wave == 0 && h < n_iter) out[h] = CoarseHyp{0, -1, 0.0, 0.0};
        return;
    }
    if (wave == 0) {
        double T[12];
        const bool ok = (h < n_iter) && sample_T(pts, C, (uint32_t)h, seed, T);
        double eta = 0.0;
        float f[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1.0f, -1.0f};
        bool sure = true;  // degenerate / out-of-range hypotheses do not veto the fast path
        if (ok) {
            double tmax = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double tr = (T[4 * r + 3] + ((T[4 * r] * stats->cs[0] + T[4 * r + 1] * stats->cs[1]) + T[4 * r + 2] * stats->cs[2])) -
                                  stats->cq[r];
                f[9 + r] = (float)tr;
                tmax = fmax(tmax, fabs(tr));
                f[3 * r] = (float)T[4 * r];
                f[3 * r + 1] = (float)T[4 * r + 1];
                f[3 * r + 2] = (float)T[4 * r + 2];
            }
            const double delta = 5.9604644775390625e-08 * (32.0 * stats->M + 8.0 * tmax);  // u = 2^-24
            eta = 1.7321 * delta;
            // certain / possible inlier thresholds on the fp32 squared distance (directed 1e-6 slack >> fp32 rounding)
            if (max_dist > 0.0) {
                const double lo = fmax(0.0, max_dist - eta), hi = max_dist + eta;
                f[12] = (float)(lo * lo * (1.0 - 1e-6));
                f[13] = (float)(hi * hi * (1.0 + 1e-6));
            }
            // |d~ component| <= sqrt(3) M + |t'| + M, so d~^2 <= 3 (3 M + |t'|)^2: if even that is below the
            // "certain inlier" threshold (the reference calls RANSAC with max_dist = 10000 m), no test is needed
            const double reach = 3.0 * stats->M + tmax;
            sure = (double)f[12] > 3.0 * reach * reach * 1.001;
        }
        const bool all_sure = __all(sure);
        if (lane == 0) all_inliers_s = all_sure ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 14; ++k) hyp_f[k][lane] = f[k];
        hyp_eta[lane] = eta;
        hyp_live[lane] = ok ? 1 : 0;
    }
    __syncthreads();
    float R[9], tp[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = hyp_f[k][lane];
#pragma unroll
    for (int k = 0; k < 3; ++k) tp[k] = hyp_f[9 + k][lane];
    const float Lf = hyp_f[12][lane], Hf = hyp_f[13][lane];
    const bool all_in = all_inliers_s != 0;  // block-uniform
    int n_lo = 0, n_hi = 0;
    double E_lo = 0.0, E_hi = 0.0;
    const int64_t nchunks = (C + COARSE_CHUNK - 1) / COARSE_CHUNK;  // 768 floats per chunk = 64 lanes x 3 float4
    float4 pre[3];
    auto fetch = [&](int64_t c) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int64_t e = c * (COARSE_CHUNK * 6) + (int64_t)(j * 64 + lane) * 4;
            // pts32 is allocated with 4 floats of slack: a float4 that STARTS inside the data may over-read
            pre[j] = (e < C * 6) ? *reinterpret_cast<const float4*>(pts32 + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(&lbuf[wave][b][(j * 64 + lane) * 4]) = pre[j];
    };
    int b = 0;
    if (wave < nchunks) {
        fetch(wave);
        stash(0);
    }
    for (int64_t c = wave; c < nchunks; c += COARSE_WAVES, b ^= 1) {
        if (c + COARSE_WAVES < nchunks) fetch(c + COARSE_WAVES);
        __builtin_amdgcn_wave_barrier();
        const int64_t base = c * COARSE_CHUNK;
        const int cnt = (int)min((int64_t)COARSE_CHUNK, C - base);
        const float* lp = lbuf[wave][b];
        float e_lo = 0.f, e_hi = 0.f;
        auto score = [&](const float* p, float& d2) {  // p: LDS broadcast address
            const float dx = fmaf(R[0], p[0], fmaf(R[1], p[1], fmaf(R[2], p[2], tp[0]))) - p[3];
            const float dy = fmaf(R[3], p[0], fmaf(R[4], p[1], fmaf(R[5], p[2], tp[1]))) - p[4];
            const float dz = fmaf(R[6], p[0], fmaf(R[7], p[1], fmaf(R[8], p[2], tp[2]))) - p[5];
            d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        };
        if (all_in) {
            // every point is a certain inlier: only the sum is needed (4 partial sums for ILP)
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int i = 0;
            for (; i + 4 <= cnt; i += 4) {
                float d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) score(lp + 6 * (i + u), d[u]);
                a0 += d[0]; a1 += d[1]; a2 += d[2]; a3 += d[3];
            }
            for (; i < cnt; ++i) {
                float d;
                score(lp + 6 * i, d);
                a0 += d;
            }
            e_lo = e_hi = (a0 + a1) + (a2 + a3);
            n_lo += cnt;
            n_hi += cnt;
        } else {
            auto tally = [&](float d2) {
                const bool in_lo = d2 < Lf, in_hi = d2 < Hf;
                n_lo += in_lo ? 1 : 0;
                n_hi += in_hi ? 1 : 0;
                e_lo += in_lo ? d2 : 0.f;
                e_hi += in_hi ? d2 : 0.f;
            };
            int i = 0;
            for (; i + 4 <= cnt; i += 4) {
                float d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) score(lp + 6 * (i + u), d[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) tally(d[u]);
            }
            for (; i < cnt; ++i) {
                float d;
                score(lp + 6 * i, d);
                tally(d);
            }
        }
        E_lo += (double)e_lo;
        E_hi += (double)e_hi;
        __builtin_amdgcn_wave_barrier();
        if (c + COARSE_WAVES < nchunks) stash(b ^ 1);
    }
    part_n[wave][0][lane] = n_lo;
    part_n[wave][1][lane] = n_hi;
    part_e[wave][0][lane] = E_lo;
    part_e[wave][1][lane] = E_hi;
    __syncthreads();
    if (wave != 0) return;
    n_lo = n_hi = 0;
    E_lo = E_hi = 0.0;
#pragma unroll
    for (int w = 0; w < COARSE_WAVES; ++w) {  // fixed order: deterministic
        n_lo += part_n[w][0][lane];
        n_hi += part_n[w][1][lane];
        E_lo += part_e[w][0][lane];
        E_hi += part_e[w][1][lane];
    }
    const bool live = hyp_live[lane] != 0;
    const double eta = hyp_eta[lane];
    if (h < n_iter) {
        CoarseHyp o;
        if (!live) {
            o = CoarseHyp{0, -1, 0.0, 0.0};
        } else {
            o.n_lo = n_lo;
            o.n_hi = n_hi;
            o.r_lo = (n_lo > 0) ? fmax(0.0, sqrt(E_lo / (double)n_lo) * (1.0 - 1e-4) - 3.0 * eta) : 0.0;
            o.r_hi = (n_hi > 0) ? sqrt(E_hi / (double)n_hi) * (1.0 + 1e-4) + eta : 1.7976931348623157e308;
        }
        out[h] = o;
    }
    // F* = max over hypotheses of the certain inlier count (one atomic per workgroup)
    int f = (live && h < n_iter) ? n_lo : 0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) f = max(f, __shfl_xor(f, off));
    if (lane == 0 && f > 0) atomicMax(&sel->F, f);
}
__global__ __launch_bounds__(512) void ransac_coarse_kernel(const double* __restrict__ pts, const float* __restrict__ pts32,
                                                            const RansacStats* __restrict__ stats,
                                                            const int64_t* __restrict__ count_dev, int64_t c_max,
                                                            double max_dist, int32_t n_iter, uint64_t seed,
                                                            CoarseHyp* __restrict__ out, SelectState* __restrict__ sel) {
    ransac_coarse_block((int)blockIdx.x, pts, pts32, stats, count_dev, c_max, max_dist, n_iter, seed, out, sel);
}
// Fused chain: the same pass from a SMALL grid that walks the blocks of 64 hypotheses.  In the regime the reference runs (max distance
// 10 000 m: every hypothesis bounded in closed form) the pass has nothing to do, and 782 workgroups of 512 threads + 66 KB of LDS that
// only look at a flag took 5 us alone and 60 us beside a coarse kernel of the matcher, whose workgroups own their compute units
// (profiles/r06_trace_pipe_d2.txt); now it is as many workgroups as fit at once.  Where it does run, its LAST workgroup (a ticket)
// recomputes R* from the bounds it wrote -- ransac_select_rmin_kernel's pass -- with the final F.
__global__ __launch_bounds__(512) void ransac_coarse_loop_kernel(const double* __restrict__ pts, const float* __restrict__ pts32,
                                                                 const RansacStats* __restrict__ stats,
                                                                 const int64_t* __restrict__ count_dev, int64_t c_max,
                                                                 double max_dist, int32_t n_iter, uint64_t seed,
                                                                 CoarseHyp* __restrict__ out, SelectState* __restrict__ sel) {
    __shared__ int last_s;
    __shared__ double rmin_s[8];
    if (sel->unsure == 0) return;
    const int nblocks = (n_iter + 63) / 64;
    for (int hb = (int)blockIdx.x; hb < nblocks; hb += (int)gridDim.x) {
        __syncthreads();   // (the block's LDS tables are reused)
        ransac_coarse_block(hb, pts, pts32, stats, count_dev, c_max, max_dist, n_iter, seed, out, sel);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last_s = (atomicAdd(&sel->done, 1) == (int)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    const int F = __hip_atomic_load(&sel->F, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double r = 1.7976931348623157e308;
    for (int h = threadIdx.x; h < n_iter; h += 512) {
        const CoarseHyp c = out[h];
        if (c.n_hi >= 0 && c.n_lo == F && c.n_hi == F) r = fmin(r, c.r_hi);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r = fmin(r, __shfl_xor(r, off));
    if ((threadIdx.x & 63) == 0) rmin_s[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 8; ++w) r = fmin(r, rmin_s[w]);
        sel->Rbits = r < 1.7976931348623157e308 ? (unsigned long long)__double_as_longlong(r) : 0x7FF0000000000000ull;
    }
}

// which hypotheses can still be the exact winner?
//   F* = max n_lo (some hypothesis certainly has that many inliers); a winner needs n_hi >= F*.
//   R* = min r_hi over hypotheses whose count is certainly F* (n_lo == n_hi == F*); a hypothesis that
//        can at best tie on the count (n_hi == F*) also needs r_lo <= R*.
__global__ __launch_bounds__(256) void ransac_select_rmin_kernel(const CoarseHyp* __restrict__ hyps, int32_t n_iter,
                                                                 SelectState* __restrict__ sel) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    const int F = sel->F;
    double r = 1.7976931348623157e308;
    if (h < n_iter) {
        const CoarseHyp c = hyps[h];
        if (c.n_hi >= 0 && c.n_lo == F && c.n_hi == F) r = c.r_hi;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) r = fmin(r, __shfl_xor(r, off));
    if ((threadIdx.x & 63) == 0 && r < 1.7976931348623157e308)
        atomicMin(&sel->Rbits, (unsigned long long)__double_as_longlong(r));  // r >= 0: bit order == value order
}

__global__ __launch_bounds__(256) void ransac_select_list_kernel(const CoarseHyp* __restrict__ hyps, int32_t n_iter,
                                                                 SelectState* __restrict__ sel, int32_t* __restrict__ list) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= n_iter) return;
    const int F = sel->F;
    const double Rs = __longlong_as_double((long long)sel->Rbits);
    const CoarseHyp c = hyps[h];
    const bool cand = c.n_hi > 0 && c.n_hi >= F && (c.n_hi > F || c.r_lo <= Rs);
    if (cand) {
        const int slot = atomicAdd(&sel->count, 1);
        if (slot < CAND_MAX) list[slot] = h;
        else sel->overflow = 1;
    }
}

// Exact scoring of ONE candidate per wavefront.  The 64 lanes evaluate the residuals of a 1024-point
// chunk in parallel (same fp64 expression as the oracle), then the chunk is folded into (good, e2) in
// correspondence order by a sequential loop over LDS -- the oracle's accumulation order -- so a
// candidate costs ~C * (1/64 * 31 + 1) fp64 ops of latency instead of C * 31.
constexpr int EXACT_CHUNK = 1024;
// the oracle's score of hypothesis h by one wavefront (see ransac_exact_list_kernel below): every lane returns the same HypScore
__device__ __forceinline__ HypScore ransac_exact_one(const double* __restrict__ pts, int64_t C, double max_d2, int32_t h, uint64_t seed,
                                                     double (&d2s)[EXACT_CHUNK]) {
    const int lane = threadIdx.x & 63;
    HypScore res;
    res.fit = 0.0;
    res.rmse = 0.0;
    res.hyp = -1;
    double T[12];
    if (!sample_T(pts, C, (uint32_t)h, seed, T)) return res;   // wave-uniform
    int64_t good = 0;
    double e2 = 0.0;
    for (int64_t base = 0; base < C; base += EXACT_CHUNK) {
        const int cnt = (int)min((int64_t)EXACT_CHUNK, C - base);
        __builtin_amdgcn_wave_barrier();
        int mine = 0;
        for (int i = lane; i < cnt; i += 64) {
            const double* p = pts + 6 * (base + i);
            const double d2 = err2(T, p[0], p[1], p[2], p[3], p[4], p[5]);
            const bool in = d2 < max_d2;
            mine += in ? 1 : 0;
            d2s[i] = in ? d2 : 0.0;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
        good += mine;
        __builtin_amdgcn_wave_barrier();
        int i = 0;
        if (cnt >= 32) {
            double va[16], vb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) va[u] = d2s[u];
            for (; i + 48 <= cnt; i += 32) {
#pragma unroll
                for (int u = 0; u < 16; ++u) vb[u] = d2s[i + 16 + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) e2 = e2 + va[u];
#pragma unroll
                for (int u = 0; u < 16; ++u) va[u] = d2s[i + 32 + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) e2 = e2 + vb[u];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) e2 = e2 + va[u];
            i += 16;
        }
        for (; i < cnt; ++i) e2 = e2 + d2s[i];
    }
    if (good > 0) {
        res.fit = (double)good / (double)C;
        res.rmse = sqrt(e2 / (double)good);
        res.hyp = h;
    }
    return res;
}

// Fused chain: ransac_select_list_kernel + ransac_exact_list_kernel in one launch and without the list -- the wave that owns 64
// hypotheses tests them against (F*, R*) and scores the ones that can still win itself, one after the other (in the reference's
// regime ONE hypothesis of 50 000 survives the closed-form bound).  No candidate cap, hence no overflow pass behind it.
__global__ __launch_bounds__(64) void ransac_select_exact_kernel(const double* __restrict__ pts, const int64_t* __restrict__ count_dev,
                                                                 int64_t c_max, double max_d2, int32_t n_iter, uint64_t seed,
                                                                 const CoarseHyp* __restrict__ hyps, SelectState* __restrict__ sel,
                                                                 HypScore* __restrict__ block_best) {
    __shared__ __attribute__((aligned(16))) double d2s[EXACT_CHUNK];
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    const int lane = threadIdx.x;
    const int h = (int)blockIdx.x * 64 + lane;
    const int F = sel->F;
    const double Rs = __longlong_as_double((long long)sel->Rbits);
    bool cand = false;
    if (h < n_iter && C >= 3) {
        const CoarseHyp c = hyps[h];
        cand = c.n_hi > 0 && c.n_hi >= F && (c.n_hi > F || c.r_lo <= Rs);
    }
    unsigned long long todo = __ballot(cand);
    HypScore best;
    best.fit = 0.0;
    best.rmse = 0.0;
    best.hyp = -1;
    if (todo != 0ull && lane == 0) atomicAdd(&sel->count, (int)__popcll(todo));   // (vfm_debug_ransac_counts)
    while (todo != 0ull) {   // ascending hypothesis index: wave-uniform
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const HypScore r = ransac_exact_one(pts, C, max_d2, (int)blockIdx.x * 64 + l, seed, d2s);
        if (better(r.fit, r.rmse, r.hyp, best.fit, best.rmse, best.hyp)) best = r;
    }
    if (lane == 0) block_best[blockIdx.x] = best;
}

// Fused chain: ransac_final_kernel + ransac_mask_kernel in one workgroup (the mask is 10^4 residuals).
__global__ __launch_bounds__(1024) void ransac_final_mask_kernel(const double* __restrict__ pts, const int64_t* __restrict__ count_dev,
                                                                 int64_t c_max, double max_d2, uint64_t seed,
                                                                 const HypScore* __restrict__ block_best, int nblocks,
                                                                 double* __restrict__ T_out, double* __restrict__ fitness_out,
                                                                 double* __restrict__ rmse_out, int32_t* __restrict__ best_hyp_out,
                                                                 uint8_t* __restrict__ mask) {
    __shared__ double sf[16], sr[16];
    __shared__ int sh[16];
    __shared__ double Ts[12];
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    double fit = 0.0, rmse = 0.0;
    int hyp = -1;
    for (int b = threadIdx.x; b < nblocks; b += 1024) {
        const HypScore s = block_best[b];
        if (better(s.fit, s.rmse, s.hyp, fit, rmse, hyp)) {
            fit = s.fit;
            rmse = s.rmse;
            hyp = s.hyp;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double of = __shfl_xor(fit, off), orr = __shfl_xor(rmse, off);
        const int oh = __shfl_xor(hyp, off);
        if (better(of, orr, oh, fit, rmse, hyp)) {
            fit = of;
            rmse = orr;
            hyp = oh;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        sf[threadIdx.x >> 6] = fit;
        sr[threadIdx.x >> 6] = rmse;
        sh[threadIdx.x >> 6] = hyp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (better(sf[w], sr[w], sh[w], sf[0], sr[0], sh[0])) {
                sf[0] = sf[w];
                sr[0] = sr[w];
                sh[0] = sh[w];
            }
        double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        if (sh[0] >= 0) sample_T(pts, C, (uint32_t)sh[0], seed, T);
        for (int k = 0; k < 12; ++k) {
            T_out[k] = T[k];
            Ts[k] = T[k];
        }
        T_out[12] = 0.0;
        T_out[13] = 0.0;
        T_out[14] = 0.0;
        T_out[15] = 1.0;
        *fitness_out = (sh[0] >= 0) ? sf[0] : 0.0;
        *rmse_out = (sh[0] >= 0) ? sr[0] : 0.0;
        *best_hyp_out = sh[0];
    }
    __syncthreads();
    if (!mask) return;
    const bool have = sh[0] >= 0;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = Ts[k];
    for (int64_t i = threadIdx.x; i < c_max; i += 1024) {
        uint8_t m = 0;
        if (i < C && have) {
            const double* p = pts + 6 * i;
            m = (err2(T, p[0], p[1], p[2], p[3], p[4], p[5]) < max_d2) ? 1 : 0;
        }
        mask[i] = m;
    }
}

__global__ __launch_bounds__(64) void ransac_exact_list_kernel(const double* __restrict__ pts,
                                                               const int64_t* __restrict__ count_dev, int64_t c_max,
                                                               double max_d2, int32_t n_iter, uint64_t seed,
                                                               const int32_t* __restrict__ list,
                                                               const SelectState* __restrict__ sel,
                                                               HypScore* __restrict__ best_out) {
    __shared__ __attribute__((aligned(16))) double d2s[EXACT_CHUNK];
    const int64_t C = count_dev ? min(*count_dev, c_max) : c_max;
    const int lane = threadIdx.x;
    HypScore res;
    res.fit = 0.0;
    res.rmse = 0.0;
    res.hyp = -1;
    const int ncand = sel->overflow ? 0 : min(sel->count, CAND_MAX);
    if ((int)blockIdx.x < ncand && C >= 3) {
        const int32_t h = list[blockIdx.x];
        double T[12];
        if (sample_T(pts, C, (uint32_t)h, seed, T)) {  // wave-uniform
            int64_t good = 0;
            double e2 = 0.0;
            for (int64_t base = 0; base < C; base += EXACT_CHUNK) {
                const int cnt = (int)min((int64_t)EXACT_CHUNK, C - base);
                __builtin_amdgcn_wave_barrier();
                // parallel phase: residual, inlier test and inlier count (a count has no order);
                // LDS receives "d2 if inlier else 0.0" -- adding 0.0 is the identity, so the in-order
                // fold below is bit-identical to "if (d2 < max_d2) e2 += d2"
                int mine = 0;
                for (int i = lane; i < cnt; i += 64) {
                    const double* p = pts + 6 * (base + i);
                    const double d2 = err2(T, p[0], p[1], p[2], p[3], p[4], p[5]);
                    const bool in = d2 < max_d2;
                    mine += in ? 1 : 0;
                    d2s[i] = in ? d2 : 0.0;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
                good += mine;
                __builtin_amdgcn_wave_barrier();
                // sequential phase: the oracle's accumulation order (every lane runs the same chain)
                // two register batches: the LDS reads of the next 16 values are in flight while the
                // dependent add chain consumes the current 16 (the sum stays strictly in order)
                int i = 0;
                if (cnt >= 32) {
                    double va[16], vb[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) va[u] = d2s[u];
                    for (; i + 48 <= cnt; i += 32) {
#pragma unroll
                        for (int u = 0; u < 16; ++u) vb[u] = d2s[i + 16 + u];
#pragma unroll
                        for (int u = 0; u < 16; ++u) e2 = e2 + va[u];
#pragma unroll
                        for (int u = 0; u < 16; ++u) va[u] = d2s[i + 32 + u];
#pragma unroll
                        for (int u = 0; u < 16; ++u) e2 = e2 + vb[u];
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) e2 = e2 + va[u];  // va holds d2s[i .. i + 15]
                    i += 16;
                }
                for (; i < cnt; ++i) e2 = e2 + d2s[i];
            }
            if (good > 0) {
                res.fit = (double)good / (double)C;
                res.rmse = sqrt(e2 / (double)good);
                res.hyp = h;
            }
        }
    }
    if (lane == 0) best_out[blockIdx.x] = res;
}

struct RansacWs {
    double* pts;
    float* pts32;
    RansacStats* stats;
    CoarseHyp* hyps;
    int32_t* list;
    SelectState* sel;
    HypScore* block_best;  // [0, CAND_MAX): one per candidate, [CAND_MAX, CAND_MAX + nblocks): full fallback pass
    size_t bytes;
};
inline RansacWs carve_ransac(void* p, int64_t c_max, int32_t n_iter) {
    VfmCarver c(p);
    RansacWs w;
    w.pts = c.take<double>((size_t)(c_max > 0 ? c_max : 1) * 6);
    w.pts32 = c.take<float>((size_t)(c_max > 0 ? c_max : 1) * 6 + 4);
    w.stats = c.take<RansacStats>(1);
    w.hyps = c.take<CoarseHyp>((size_t)n_iter);
    w.list = c.take<int32_t>(CAND_MAX);
    w.sel = c.take<SelectState>(1);
    w.block_best = c.take<HypScore>((size_t)(n_iter + 63) / 64 + CAND_MAX + 1);
    w.bytes = c.used();
    return w;
}


}  // namespace


// tests / bench: what the last vfm_ransac_corr in `ws` did -- out_host[0] = hypotheses scored in the oracle's fp64 arithmetic from the
// candidate list, [1] = 1 if the list overflowed (then every hypothesis was scored in fp64), [2] = 1 if the point-wise fp32 pass was
// needed (some hypothesis not provably all-inlier).  Synchronises the device.
VFM_EXPORT int vfm_debug_ransac_counts(const void* ws, int64_t c_max, int32_t n_iter, int32_t* out_host) {
    VFM_CHECK_ARG(ws && out_host && c_max >= 0 && n_iter > 0, "ransac_counts: bad arguments");
    RansacWs w = carve_ransac(const_cast<void*>(ws), c_max, n_iter);
    SelectState h;
    VFM_CHECK_HIP(hipMemcpy(&h, w.sel, sizeof(h), hipMemcpyDeviceToHost));
    out_host[0] = h.count < CAND_MAX ? h.count : CAND_MAX;
    out_host[1] = h.overflow;
    out_host[2] = h.unsure;
    return VFM_OK;
}

VFM_EXPORT size_t vfm_ransac_workspace_bytes(int64_t c_max, int32_t n_iter) { return carve_ransac(nullptr, c_max, n_iter).bytes; }

namespace {
int ransac_corr_impl(const double* src, int64_t ns, const double* tgt, int64_t nt, const int32_t* corres, const int64_t* count_dev,
                     int64_t c_max, double max_dist, int32_t n_iter, uint64_t seed, double* T_out, double* fitness_out,
                     double* rmse_out, uint8_t* inlier_mask, int32_t* best_hyp_out, int32_t* bad_out, void* ws, size_t ws_bytes,
                     vfm_stream_t stream);
}
VFM_EXPORT int vfm_ransac_corr(const double* src, const double* tgt, const int32_t* corres, const int64_t* count_dev,
                               int64_t c_max, double max_dist, int32_t n_iter, uint64_t seed, double* T_out,
                               double* fitness_out, double* rmse_out, uint8_t* inlier_mask, int32_t* best_hyp_out,
                               void* ws, size_t ws_bytes, vfm_stream_t stream) {
    return ransac_corr_impl(src, 0, tgt, 0, corres, count_dev, c_max, max_dist, n_iter, seed, T_out, fitness_out, rmse_out, inlier_mask,
                            best_hyp_out, nullptr, ws, ws_bytes, stream);
}
VFM_EXPORT int vfm_ransac_corr_bounded(const double* src, int64_t ns, const double* tgt, int64_t nt, const int32_t* corres,
                                       const int64_t* count_dev, int64_t c_max, double max_dist, int32_t n_iter, uint64_t seed,
                                       double* T_out, double* fitness_out, double* rmse_out, uint8_t* inlier_mask,
                                       int32_t* best_hyp_out, int32_t* bad_out, void* ws, size_t ws_bytes, vfm_stream_t stream) {
    VFM_CHECK_ARG(bad_out && ns > 0 && nt > 0, "ransac: bad_out / cloud sizes");
    return ransac_corr_impl(src, ns, tgt, nt, corres, count_dev, c_max, max_dist, n_iter, seed, T_out, fitness_out, rmse_out, inlier_mask,
                            best_hyp_out, bad_out, ws, ws_bytes, stream);
}
namespace {
int ransac_corr_impl(const double* src, int64_t ns, const double* tgt, int64_t nt, const int32_t* corres, const int64_t* count_dev,
                     int64_t c_max, double max_dist, int32_t n_iter, uint64_t seed, double* T_out, double* fitness_out,
                     double* rmse_out, uint8_t* inlier_mask, int32_t* best_hyp_out, int32_t* bad_out, void* ws, size_t ws_bytes,
                     vfm_stream_t stream) {
    VFM_CHECK_ARG(src && tgt && corres && T_out && fitness_out && rmse_out && best_hyp_out && ws, "ransac: null pointer");
    VFM_CHECK_ARG(c_max >= 0 && n_iter > 0, "ransac: bad sizes (c_max=%lld n_iter=%d)", (long long)c_max, n_iter);
    if (ws_bytes < vfm_ransac_workspace_bytes(c_max, n_iter)) return vfm_fail(VFM_EWORKSPACE, "ransac: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    RansacWs w = carve_ransac(ws, c_max, n_iter);
    // Open3D returns a default RegistrationResult when max_correspondence_distance <= 0:
    // a non-positive squared threshold admits no inlier, which yields exactly that here.
    const double max_d2 = (max_dist > 0.0) ? max_dist * max_dist : -1.0;
    const int nblocks = (n_iter + 63) / 64;
    if (vfm_cfg().ransac_fused == 1 && !vfm_cfg().ransac_exact_only && c_max > 0) {
        // chain 1 (round 6, measured and NOT the default: 0.195 ms against 0.187 at 10^4 correspondences -- the stage is its kernels'
        // own dependent chains, not its launch boundaries: tools/time_ransac.py): 5 launches for the 11 below, same winner
        static std::atomic<int> ncu{0};
        int cus = ncu.load(std::memory_order_relaxed);
        if (cus == 0) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            ncu.store(cus, std::memory_order_relaxed);
        }
        hipLaunchKernelGGL(ransac_prepare_kernel, dim3(1), dim3(1024), 0, st, src, tgt, corres, count_dev, c_max, w.pts, ns, nt, bad_out,
                           w.stats, w.pts32, w.sel);
        hipLaunchKernelGGL(ransac_moment_kernel, dim3(nblocks), dim3(64), 0, st, w.pts, w.stats, count_dev, c_max, max_d2, n_iter, seed,
                           w.hyps, w.sel, 1);
        hipLaunchKernelGGL(ransac_coarse_loop_kernel, dim3(nblocks < cus ? nblocks : cus), dim3(64 * COARSE_WAVES), 0, st, w.pts, w.pts32,
                           w.stats, count_dev, c_max, max_dist, n_iter, seed, w.hyps, w.sel);
        hipLaunchKernelGGL(ransac_select_exact_kernel, dim3(nblocks), dim3(64), 0, st, w.pts, count_dev, c_max, max_d2, n_iter, seed,
                           w.hyps, w.sel, w.block_best + CAND_MAX);
        VFM_CHECK_LAUNCH("ransac fused chain");
        hipLaunchKernelGGL(ransac_final_mask_kernel, dim3(1), dim3(1024), 0, st, w.pts, count_dev, c_max, max_d2, seed,
                           w.block_best + CAND_MAX, nblocks, T_out, fitness_out, rmse_out, best_hyp_out, inlier_mask);
        VFM_CHECK_LAUNCH("ransac_final_mask_kernel");
        return VFM_OK;
    }
    if (c_max > 0) {
        hipLaunchKernelGGL(ransac_gather_kernel, dim3((unsigned)((c_max + 255) / 256)), dim3(256), 0, st, src, tgt, corres,
                           count_dev, c_max, w.pts, ns, nt, bad_out);
        VFM_CHECK_LAUNCH("ransac_gather_kernel");
    }
    if (vfm_cfg().ransac_exact_only) {
        // reference-order fp64 scoring of every hypothesis (A/B switch, and the semantics the
        // two-level path below must reproduce bit for bit)
        hipLaunchKernelGGL(ransac_score_kernel, dim3(nblocks), dim3(64), 0, st, w.pts, count_dev, c_max, max_d2, n_iter, seed,
                           (const int32_t*)nullptr, w.block_best);
        VFM_CHECK_LAUNCH("ransac_score_kernel");
        hipLaunchKernelGGL(ransac_final_kernel, dim3(1), dim3(256), 0, st, w.pts, count_dev, c_max, seed, w.block_best,
                           nblocks, T_out, fitness_out, rmse_out, best_hyp_out);
    } else {
        // coarse bounds (closed form from the stream's moments when every point is provably an inlier,
        // else the point-wise fp32 pass) -> candidates -> exact fp64 on the candidates (or on everything
        // if the candidate list overflowed)
        const unsigned gsel = (unsigned)((n_iter + 255) / 256);
        if (vfm_cfg().ransac_fused == 2) {
            // chain 2 (round 6, default): round 5's kernels where one of them alone is the faster one (the parallel gather, the candidate
            // list + one wave per candidate), minus the launches that only hand a flag on: the select state is reset by the centring
            // kernel, R* comes out of the moment pass (F = C is known there), the point-wise pass is a small grid that looks at the flag,
            // final + mask are one workgroup.  8 launches for 11.
            int cus = 256, dev = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            hipLaunchKernelGGL(ransac_center_kernel, dim3(1), dim3(1024), 0, st, w.pts, count_dev, c_max, w.stats, w.pts32, w.sel);
            hipLaunchKernelGGL(ransac_moment_kernel, dim3(nblocks), dim3(64), 0, st, w.pts, w.stats, count_dev, c_max, max_d2,
                               n_iter, seed, w.hyps, w.sel, 1);
            hipLaunchKernelGGL(ransac_coarse_loop_kernel, dim3(nblocks < cus ? nblocks : cus), dim3(64 * COARSE_WAVES), 0, st, w.pts, w.pts32,
                               w.stats, count_dev, c_max, max_dist, n_iter, seed, w.hyps, w.sel);
            hipLaunchKernelGGL(ransac_select_list_kernel, dim3(gsel), dim3(256), 0, st, w.hyps, n_iter, w.sel, w.list);
            hipLaunchKernelGGL(ransac_exact_list_kernel, dim3(CAND_MAX), dim3(64), 0, st, w.pts, count_dev, c_max, max_d2, n_iter,
                               seed, w.list, w.sel, w.block_best);
            hipLaunchKernelGGL(ransac_score_kernel, dim3(nblocks), dim3(64), 0, st, w.pts, count_dev, c_max, max_d2, n_iter, seed,
                               &w.sel->overflow, w.block_best + CAND_MAX);
            VFM_CHECK_LAUNCH("ransac coarse/select/score kernels");
            hipLaunchKernelGGL(ransac_final_mask_kernel, dim3(1), dim3(1024), 0, st, w.pts, count_dev, c_max, max_d2, seed, w.block_best,
                               CAND_MAX + nblocks, T_out, fitness_out, rmse_out, best_hyp_out, c_max > 0 ? inlier_mask : (uint8_t*)nullptr);
            VFM_CHECK_LAUNCH("ransac_final_mask_kernel");
            return VFM_OK;
        }
        hipLaunchKernelGGL(ransac_sel_init_kernel, dim3(1), dim3(1), 0, st, w.sel);
        hipLaunchKernelGGL(ransac_center_kernel, dim3(1), dim3(1024), 0, st, w.pts, count_dev, c_max, w.stats, w.pts32, (SelectState*)nullptr);
        hipLaunchKernelGGL(ransac_moment_kernel, dim3(nblocks), dim3(64), 0, st, w.pts, w.stats, count_dev, c_max, max_d2,
                           n_iter, seed, w.hyps, w.sel, 0);
        hipLaunchKernelGGL(ransac_coarse_kernel, dim3(nblocks), dim3(64 * COARSE_WAVES), 0, st, w.pts, w.pts32, w.stats, count_dev, c_max,
                           max_dist, n_iter, seed, w.hyps, w.sel);
        hipLaunchKernelGGL(ransac_select_rmin_kernel, dim3(gsel), dim3(256), 0, st, w.hyps, n_iter, w.sel);
        hipLaunchKernelGGL(ransac_select_list_kernel, dim3(gsel), dim3(256), 0, st, w.hyps, n_iter, w.sel, w.list);
        hipLaunchKernelGGL(ransac_exact_list_kernel, dim3(CAND_MAX), dim3(64), 0, st, w.pts, count_dev, c_max, max_d2, n_iter,
                           seed, w.list, w.sel, w.block_best);
        hipLaunchKernelGGL(ransac_score_kernel, dim3(nblocks), dim3(64), 0, st, w.pts, count_dev, c_max, max_d2, n_iter, seed,
                           &w.sel->overflow, w.block_best + CAND_MAX);
        VFM_CHECK_LAUNCH("ransac coarse/select/score kernels");
        hipLaunchKernelGGL(ransac_final_kernel, dim3(1), dim3(256), 0, st, w.pts, count_dev, c_max, seed, w.block_best,
                           CAND_MAX + nblocks, T_out, fitness_out, rmse_out, best_hyp_out);
    }
    VFM_CHECK_LAUNCH("ransac_final_kernel");
    if (inlier_mask && c_max > 0) {
        hipLaunchKernelGGL(ransac_mask_kernel, dim3((unsigned)((c_max + 255) / 256)), dim3(256), 0, st, w.pts, count_dev,
                           c_max, max_d2, T_out, best_hyp_out, inlier_mask);
        VFM_CHECK_LAUNCH("ransac_mask_kernel");
    }
    return VFM_OK;
}
}  // namespace

VFM_EXPORT int vfm_kabsch_batched(const double* A, const double* B, const double* w, int64_t b, int64_t n,
                                  double denom_eps, double* T_out, int32_t* valid, vfm_stream_t stream) {
    VFM_CHECK_ARG(A && B && T_out && b >= 0 && n >= 1, "kabsch: bad arguments");
    if (b == 0) return VFM_OK;
    hipLaunchKernelGGL(kabsch_batched_kernel, dim3((unsigned)((b + 63) / 64)), dim3(64), 0, (hipStream_t)stream, A, B, w, b,
                       n, denom_eps, T_out, valid);
    VFM_CHECK_LAUNCH("kabsch_batched_kernel");
    return VFM_OK;
}
