// vit.hip -- DINOv2 ViT-S/14 (+ FeatUp ChannelNorm) forward on MI355X (gfx950), row A1.
//
// Replaces `self.model.model(torch_image)` of image_features.py:101 together with the transform of
// image_features.py:67-77 (ToTensor, bilinear Resize to 224 x 14*pw without antialias, ImageNet
// Normalize).  Architecture as in facebookresearch/dinov2 `vit_small(patch_size=14)` + FeatUp's
// `ChannelNorm` (SURVEY.md section 8 "A1 detail"): conv patch embed, cls token, (pre-interpolated)
// position embedding, 12 pre-norm blocks with LayerScale, final LayerNorm, cls dropped, LayerNorm
// over channels.
//
// MI355X design
//   * every GEMM operand (activations and weights) is kept in HBM as fp16 "fragment tiles":
//     unit (rowtile, kstep, half, row%32) = 8 consecutive k of one row = exactly one lane's
//     operand of v_mfma_f32_32x32x16_f16.  A wavefront fetches an operand with ONE coalesced
//     1 KiB load straight into VGPRs: no LDS staging, no barriers in the GEMMs.
//   * fp32 residual stream; fp32 accumulation; epilogues fused (bias, exact GELU, LayerScale +
//     residual add, position embedding).
//   * GEMMs are computed "swapped" (D = W . A^T) so a lane owns 4 consecutive output channels of
//     one token: 8-byte fragment stores / 16-byte residual read-modify-writes.
//   * attention: one wavefront per 32 queries of one (image, head); scores S^T = K Q^T so the
//     softmax row lives in one lane pair (in-lane max/sum + one xor-32 exchange); P is converted
//     to the MFMA operand layout in registers with v_permlane32_swap; K / V^T fragments come
//     straight from HBM/L2 (337 tokens x 64: 43 KB per head, L2 resident).
//   * all 6 cameras are batched (6 x 352 padded tokens = 66 row tiles) to fill the chip.
// Floating point => tolerance parity (tests state it); weights are seeded-random in tests/bench.
#include <atomic>
#include "common.h"

#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {


__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// half index of element (row, k) inside a fragment-tiled matrix with `ksteps` k-steps of 16
__device__ __forceinline__ size_t frag_index(int row, int k, int ksteps) {
    const int tile = row >> 5, p = row & 31, s = k >> 4, h = (k >> 3) & 1, e = k & 7;
    return ((((size_t)tile * ksteps + s) * 2 + h) * 32 + p) * 8 + e;
}

struct Dims {
    int B, H, W;      // images
    int Hr, Wr;       // resized image (224 x 14*pw)
    int gh, gw;       // patch grid (16 x pw)
    int Np, T, Tp;    // patches, tokens (Np+1), tokens padded to a multiple of 32
    int M;            // B * Tp rows
    int D, heads, mlp, depth;
    int KP;           // patch-embed K padded to a multiple of 32 (3*14*14 = 588 -> 608: an even number of k-steps, whole stages of the LDS-tiled GEMM)
};

// ---------------------------------------------------------------------------------------------
// weights blob
// ---------------------------------------------------------------------------------------------
// Round 4: the two LayerNorms of a block are folded into the GEMMs that consume them.  With gamma, beta the
// LayerNorm's affine, W the linear layer behind it, mu / r the token's mean / reciprocal standard deviation:
//     y_n = sum_k ((x_k - mu) r gamma_k + beta_k) W_nk + b_n  =  r (sum_k x_k W'_nk  -  mu c_n) + b'_n
//     W' = W diag(gamma)  (stored as fp16 fragments),  c_n = sum_k W'_nk  (of the ROUNDED fp16 values),  b' = b + W beta
// so the QKV / fc1 GEMMs multiply the RAW residual stream (an fp16 copy the producing epilogue writes beside the fp32 one) and
// apply mu, r in their epilogue; the producers -- patch embedding, proj and fc2 epilogues, which hold the new residual values
// in registers -- leave per-token partial sums (sum x, sum x^2) per 32-channel slice in fixed slots, added in fixed order by the
// consumer: deterministic, no atomics.  87 -> 63 launches with the same number of waves per GEMM.
// The packer (vfmreg/vit.py) does the folding on the host.
enum Seg {
    SEG_PATCH_W = 0, SEG_PATCH_B, SEG_CLS_POS,
    SEG_LAYER0,  // per layer: QKV_W' QKV_B' QKV_C PROJ_W PROJ_B LS1 FC1_W' FC1_B' FC1_C FC2_W FC2_B LS2
};
constexpr int SEGS_PER_LAYER = 12;
enum LayerSeg { L_QKV_W = 0, L_QKV_B, L_QKV_C, L_PROJ_W, L_PROJ_B, L_LS1, L_FC1_W, L_FC1_B, L_FC1_C, L_FC2_W, L_FC2_B, L_LS2 };
// tail: NORM_W NORM_B CN_W CN_B

struct Layout {
    size_t off[3 + SEGS_PER_LAYER * 64 + 4];
    size_t bytes[3 + SEGS_PER_LAYER * 64 + 4];
    int count;
    size_t total;
};

inline size_t frag_bytes(int rows, int k) { return (size_t)ceil_div(rows, 32) * 32 * (size_t)ceil_div(k, 16) * 16 * 2; }

inline Layout make_layout(const vfm_vit_config* c) {
    Layout L;
    const int D = c->dim, T = c->patch_h * c->patch_w + 1;
    const int KP = ceil_div(3 * c->patch * c->patch, 32) * 32;
    int n = 0;
    size_t off = 0;
    auto add = [&](size_t b) {
        L.off[n] = off;
        L.bytes[n] = b;
        off += vfm_align_up(b, 256);
        ++n;
    };
    add(frag_bytes(D, KP));            // patch_w  fp16 frag [D][KP]
    add((size_t)D * 4);                // patch_b  fp32
    add((size_t)T * D * 4);            // cls_pos  fp32 [T][D]: row 0 = cls + pos[0], row t = pos[t]
    for (int l = 0; l < c->depth; ++l) {
        add(frag_bytes(3 * D, D)); add((size_t)3 * D * 4); add((size_t)3 * D * 4);             // qkv: W diag(ln1 gamma), b + W ln1 beta, row sums
        add(frag_bytes(D, D)); add((size_t)D * 4); add((size_t)D * 4);                         // proj, ls1
        add(frag_bytes(c->mlp_dim, D)); add((size_t)c->mlp_dim * 4); add((size_t)c->mlp_dim * 4);   // fc1: W diag(ln2 gamma), b + W ln2 beta, row sums
        add(frag_bytes(D, c->mlp_dim)); add((size_t)D * 4); add((size_t)D * 4);                // fc2, ls2
    }
    add((size_t)D * 4); add((size_t)D * 4); add((size_t)D * 4); add((size_t)D * 4);  // norm, channel norm
    L.count = n;
    L.total = off;
    return L;
}

// ---------------------------------------------------------------------------------------------
// 1. preprocessing: uint8 HWC -> bilinear resize -> normalise -> im2col fragment tiles
//    row m = b*Tp + 1 + patch (token row), k = c*196 + py*14 + px (conv weight order)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vit_preprocess_kernel(const uint8_t* __restrict__ img, Dims d,
                                                             _Float16* __restrict__ out) {
    const int ksteps = d.KP / 16;
    const int units_per_row = ksteps * 2;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)d.M * units_per_row;
    if (gid >= total) return;
    const int m = (int)(gid / units_per_row), unit = (int)(gid % units_per_row);
    const int b = m / d.Tp, t = m % d.Tp;
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.0f;
    if (t >= 1 && t <= d.Np) {
        const int patch = t - 1, pr = patch / d.gw, pc = patch % d.gw;
        const float sh = (float)d.H / (float)d.Hr, sw = (float)d.W / (float)d.Wr;
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = unit * 8 + e;
            if (k < 3 * 196) {
                const int c = k / 196, py = (k % 196) / 14, px = k % 14;
                const int oy = pr * 14 + py, ox = pc * 14 + px;
                // torch upsample_bilinear2d, align_corners=False, antialias=False
                float sy = sh * ((float)oy + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
                float sx = sw * ((float)ox + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
                int y0 = (int)sy; if (y0 > d.H - 1) y0 = d.H - 1;
                int x0 = (int)sx; if (x0 > d.W - 1) x0 = d.W - 1;
                const int y1 = y0 + (y0 < d.H - 1 ? 1 : 0), x1 = x0 + (x0 < d.W - 1 ? 1 : 0);
                const float ly = sy - (float)y0, lx = sx - (float)x0;
                const uint8_t* base = img + (size_t)b * d.H * d.W * 3;
                const float p00 = base[((size_t)y0 * d.W + x0) * 3 + c] * (1.0f / 255.0f);
                const float p01 = base[((size_t)y0 * d.W + x1) * 3 + c] * (1.0f / 255.0f);
                const float p10 = base[((size_t)y1 * d.W + x0) * 3 + c] * (1.0f / 255.0f);
                const float p11 = base[((size_t)y1 * d.W + x1) * 3 + c] * (1.0f / 255.0f);
                const float r = (1.f - ly) * ((1.f - lx) * p00 + lx * p01) + ly * ((1.f - lx) * p10 + lx * p11);
                v[e] = (_Float16)((r - mean[c]) / stdv[c]);
            }
        }
    }
    const int s = unit >> 1, h = unit & 1;
    const size_t idx = ((((size_t)(m >> 5) * ksteps + s) * 2 + h) * 32 + (m & 31)) * 8;
    *reinterpret_cast<half8*>(out + idx) = v;
}

// Round 5 (the kernel above issues 32 single-byte loads per thread -- a thread owns 8 consecutive k = 8 samples of one
// channel, four taps each -- and stalls on their issue: 0.8 of its wave time, 189 us at 96 images).  Here a workgroup takes one token =
// one 14 x 14 patch: thread (py, px) reads its two source rows ONCE for all three channels -- the taps x0, x0 + 1 are six consecutive
// bytes: a dword and a short per row, four loads instead of twelve byte loads --, evaluates the three channels with the arithmetic of
// the kernel above (same expressions, same order: the same bits), leaves them in the LDS in k order (k = c 196 + py 14 + px) and the
// first KP / 8 threads write the token's fragment units, 16 bytes each.  Tokens without a patch (cls, padding) are zero rows.
__global__ __launch_bounds__(256) void vit_preprocess_patch_kernel(const uint8_t* __restrict__ img, Dims d, _Float16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) _Float16 vals[640];   // KP <= 640 halves (3 x 196 = 588 -> 608)
    const int m = blockIdx.x, tid = threadIdx.x;
    const int b = m / d.Tp, t = m - b * d.Tp;
    const int ksteps = d.KP / 16, units = ksteps * 2;
    const bool patch_row = t >= 1 && t <= d.Np;   // workgroup-uniform
    if (patch_row) {
        if (tid >= 196 && tid < 196 + 52) vals[588 + (tid - 196)] = (_Float16)0.0f;   // the k >= 588 tail of the last unit(s): 588 .. 639
        if (tid < 196) {
            const int patch = t - 1, pr = patch / d.gw, pc = patch - pr * d.gw;
            const int py = tid / 14, px = tid - py * 14;
            const float sh = (float)d.H / (float)d.Hr, sw = (float)d.W / (float)d.Wr;
            const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
            const int oy = pr * 14 + py, ox = pc * 14 + px;
            // torch upsample_bilinear2d, align_corners=False, antialias=False
            float sy = sh * ((float)oy + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
            float sx = sw * ((float)ox + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
            int y0 = (int)sy; if (y0 > d.H - 1) y0 = d.H - 1;
            int x0 = (int)sx; if (x0 > d.W - 1) x0 = d.W - 1;
            const int y1 = y0 + (y0 < d.H - 1 ? 1 : 0), x1 = x0 + (x0 < d.W - 1 ? 1 : 0);
            const float ly = sy - (float)y0, lx = sx - (float)x0;
            const uint8_t* base = img + (size_t)b * d.H * d.W * 3;
            // six bytes from pixel xs = min(x0, W - 2): pixels xs, xs + 1 (W >= 2); tap0 = pixel x0, tap1 = pixel x1
            const int xs = x0 < d.W - 1 ? x0 : d.W - 2;
            const int o0 = (x0 - xs) * 3, o1 = (x1 - xs) * 3;
            unsigned long long r0, r1;   // bytes 0 .. 5: the two pixels of each source row
            {
                const uint8_t* p0 = base + ((size_t)y0 * d.W + xs) * 3;
                const uint8_t* p1 = base + ((size_t)y1 * d.W + xs) * 3;
                unsigned a0, a1;
                unsigned short c0, c1;
                __builtin_memcpy(&a0, p0, 4); __builtin_memcpy(&c0, p0 + 4, 2);
                __builtin_memcpy(&a1, p1, 4); __builtin_memcpy(&c1, p1 + 4, 2);
                r0 = (unsigned long long)a0 | ((unsigned long long)c0 << 32);
                r1 = (unsigned long long)a1 | ((unsigned long long)c1 << 32);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float p00 = (float)(unsigned)((r0 >> (8 * (o0 + c))) & 0xffull) * (1.0f / 255.0f);
                const float p01 = (float)(unsigned)((r0 >> (8 * (o1 + c))) & 0xffull) * (1.0f / 255.0f);
                const float p10 = (float)(unsigned)((r1 >> (8 * (o0 + c))) & 0xffull) * (1.0f / 255.0f);
                const float p11 = (float)(unsigned)((r1 >> (8 * (o1 + c))) & 0xffull) * (1.0f / 255.0f);
                const float r = (1.f - ly) * ((1.f - lx) * p00 + lx * p01) + ly * ((1.f - lx) * p10 + lx * p11);
                vals[c * 196 + tid] = (_Float16)((r - mean[c]) / stdv[c]);
            }
        }
        __syncthreads();
    }
    if (tid < units) {
        half8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.0f;
        if (patch_row) v = *reinterpret_cast<const half8*>(vals + tid * 8);
        const int s = tid >> 1, h = tid & 1;
        const size_t idx = ((((size_t)(m >> 5) * ksteps + s) * 2 + h) * 32 + (m & 31)) * 8;
        *reinterpret_cast<half8*>(out + idx) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// 2. GEMM on fragment tiles.  out[m][n] = sum_k A[m][k] W[n][k]  (+ fused epilogue)
//    One wavefront per (32 tokens) x (64 output channels); swapped product D = W . A^T:
//    lane <-> token (column), registers <-> channels (rows): 4 consecutive channels per group.
// ---------------------------------------------------------------------------------------------
enum Epi { EPI_PATCH = 0, EPI_QKV = 1, EPI_RESID = 2, EPI_GELU = 3 };

struct GemmArgs {
    const uint4* A;    // activations, fragment tiles [M/32][KS][64]
    const uint4* W;    // weights, fragment tiles [N/32][KS][64]
    const float* bias; // [N]
    int M, N, KS;
    // epilogue operands
    float* x;            // residual stream [M][D] fp32 (PATCH: written, RESID: read-modify-write)
    const float* gamma;  // LayerScale [N] (RESID)
    const float* clspos; // [T][D] (PATCH)
    _Float16* out;       // fragment-tiled fp16 output (GELU: [M][N]; QKV: q,k,vT per head)
    int T, Tp, D, heads;
    _Float16* q;  // QKV outputs: per (b, head): Q frag [Tp][64], K frag [Tp][64], V^T frag [64][Tp]
    _Float16* k;
    _Float16* vt;
    int xcd_map;  // 1: token tile mt is worked on by workgroups with blockIdx % 8 == mt % 8 (see xcd_item)
    // LayerNorm folded into the GEMMs (see Seg): producers (PATCH, RESID) write xh + stats, consumers (QKV, GELU) read stats + csum
    _Float16* xh;        // fp16 fragment-tiled copy of the residual stream [M][D]
    float* stats;        // [M][D / 32][2]: (sum x, sum x^2) of the token over each 32-channel slice
    const float* csum;   // [N]: row sums of the (gamma-folded, fp16-rounded) weight
    float invD;          // 1 / D (the consumers' mean and variance)
    unsigned qt_magic;   // 2^20 / (Tp / 32) + 1: image of a token tile = (mt qt_magic) >> 20 (exact for mt < 2^16: vfm_vit_forward checks)
    unsigned long long* dbg;   // (tools only) per-workgroup start / end / placement of vit_gemm_astat_kernel, or null
    int hot_a;                 // (tools only, WRONG RESULTS; LDS-tiled kernel) bit 0: every workgroup reads token group 0 -- its A operand then hits the L2;
                               // bits 1 / 2 / 3: the residual epilogue without its loads of the stream / its stores to it / its fp16 copy
};

// XCD-consistent work mapping (round 3): workgroup b runs on XCD b % 8 (observed placement, speed only).  Every kernel of a block
// gives token tile mt to workgroups of XCD mt % 8, so a tile's activations are produced and consumed through the same 4 MiB L2
// instead of crossing the fabric at every kernel boundary: 0.742 instead of 0.774 ms for 6 x 1200 x 1600 (tools/ab_vit_xcd.py).
// item = (blockIdx >> 3) * 4 + wave counts the XCD's (tile, column tile) pairs; returns false past the end.
// (Also measured in round 3 and removed again -- both cut launches by concentrating a token tile's work in one workgroup: LayerNorm
// computed inside the QKV / fc1 GEMMs by every wave that needs it, 63 launches, 1.64 ms; the MLP of a block as one launch per
// 32-token tile with the hidden activations in the LDS, 75 launches, 1.09 ms -- 66 workgroups stream 2.4 MB of weights each
// through one compute unit's L2 port with a register ring's worth of loads in flight, where the wide GEMMs spread the same bytes
// over 256 compute units.)
__device__ __forceinline__ bool xcd_item(int mtiles, int ntiles, int wave, int& mt, int& nt) {
    const int xcd = blockIdx.x & 7;
    const int it = (blockIdx.x >> 3) * (int)(blockDim.x >> 6) + wave;
    const int lt = it / ntiles;
    nt = it - lt * ntiles;
    mt = xcd + 8 * lt;
    return mt < mtiles;
}

// exact (erf) GELU.  libm's erff is ~40 instructions per element, and the fc1 epilogue evaluates it 52 million times per 96-image
// forward -- 60 of the kernel's 136 us (round 4 profile).  Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 on erf, i.e. below the
// fp16 rounding of the value that is stored: 2^-11 relative) takes one reciprocal, one exp2 and seven FMAs.
__device__ __forceinline__ float gelu_exact(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);   // exp(-z^2); underflows to 0 for large |x|
    const float erf_abs = __builtin_fmaf(-(p * t), e, 1.0f);                     // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// ---- the epilogue of one 32 x 32 output tile (tokens m = mt * 32 + lane % 32, channels n32 * 32 ...), shared by the GEMM kernels.
// D layout: column = lane & 31 = token, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = channel in the tile.
//
// Round 5 (17 VALU + 10 SALU per MFMA, matrix pipe 13 - 19 % busy -- the GEMMs were bound by the issue of their own
// prologues and epilogues).  What the assembly of round 4's epilogue showed (124 VALU per 32 x 32 tile for 16 values per lane):
// an IEEE division (12 instructions) for the token's mean and another for its variance, rsqrtf's denormal guard, one multiply +
// subtract + multiply + add per value (the library is built with -ffp-contract=off), 64-bit VALU address chains for every load and
// store of a group, a per-LANE branch between the q | k and the V^T destinations (the test was on a pointer that held the lane's
// offset) and ~300 register moves to pair values up for the packed instructions the compiler did find.  Now:
//   * the token's statistics arrive as (a, nb) = (rstd, -rstd mean): one reciprocal of D from the host, v_rsq_f32;
//   * two packed fp32 FMAs per pair of values (y = acc a + (nb c + b'); residual: x + gamma (acc + b)), v_cvt_pk_f16_f32, one
//     8-byte store per group: ~1.5 instructions per value;
//   * the exact GELU on pairs (packed FMAs; 0.5 x + 0.5 |x| erf(|x| / sqrt 2) needs no copysign);
//   * every address = a wave-uniform base (scalar registers) + one 32-bit lane offset + an immediate;
//   * which of q | k | V^T a tile belongs to is a scalar branch.
// Floating point: the values differ from round 4's in the last bits (fused multiply-adds, multiplication by 1 / D); every GEMM kernel
// shares this code, so the bit-equality of the kernel variants stands (tests/test_gpu_vit.py), and the tolerance against the fp32
// oracle is unchanged.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat2(float v) { return f2{v, v}; }
__device__ __forceinline__ half4 to_half4(f2 lo, f2 hi) {
    half4 o;
    o[0] = (_Float16)lo[0]; o[1] = (_Float16)lo[1]; o[2] = (_Float16)hi[0]; o[3] = (_Float16)hi[1];
    return o;
}
// exact GELU of a pair (gelu_exact's formula: Abramowitz & Stegun 7.1.26), x Phi(x) = 0.5 x + 0.5 |x| erf(|x| / sqrt 2)
__device__ __forceinline__ f2 gelu2(f2 x) {
    const f2 ax = f2{fabsf(x[0]), fabsf(x[1])};
    const f2 z = ax * splat2(0.70710678118654752440f);
    const f2 den = fma2(splat2(0.3275911f), z, splat2(1.0f));
    const f2 t = f2{__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f2 p = fma2(splat2(1.061405429f), t, splat2(-1.453152027f));
    p = fma2(p, t, splat2(1.421413741f));
    p = fma2(p, t, splat2(-0.284496736f));
    p = fma2(p, t, splat2(0.254829592f));
    const f2 ea = (z * z) * splat2(-1.44269504088896340736f);
    const f2 e = f2{__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};   // exp(-z^2); underflows to 0 for large |x|
    const f2 erf_abs = fma2(-(p * t), e, splat2(1.0f));                                // erf(|x| / sqrt 2)
    return fma2(ax * splat2(0.5f), erf_abs, x * splat2(0.5f));
}
struct EpiRegs {
    float4 bias[4], aux[4], x[4];   // aux: row sums of the folded weight (QKV, fc1) / LayerScale (proj, fc2); x: residual values / cls + pos
};
// (n32u: the 32-channel tile, wave-uniform; the lane's part of every address is 16 hi bytes -- or its row of the residual stream)
template <int EPI>
__device__ __forceinline__ void epi_load(const GemmArgs& g, int m, int t, int hi, int n32u, EpiRegs& e) {
    constexpr bool CONSUMES_LN = EPI == EPI_QKV || EPI == EPI_GELU;
    const unsigned hoff = 16u * (unsigned)hi;
    const char* bias_b = reinterpret_cast<const char*>(g.bias + n32u * 32);
    const char* aux_b = reinterpret_cast<const char*>((CONSUMES_LN ? g.csum : g.gamma) + n32u * 32);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        e.bias[grp] = *reinterpret_cast<const float4*>(bias_b + (hoff + 32u * grp));
        e.aux[grp] = make_float4(0.f, 0.f, 0.f, 0.f);
        e.x[grp] = make_float4(0.f, 0.f, 0.f, 0.f);   // (padding rows t >= T stay exactly zero)
        if constexpr (CONSUMES_LN || EPI == EPI_RESID) e.aux[grp] = *reinterpret_cast<const float4*>(aux_b + (hoff + 32u * grp));
    }
    if constexpr (EPI == EPI_RESID) {
        if (t < g.T && !(g.hot_a & 2)) {
            const char* xb = reinterpret_cast<const char*>(g.x + n32u * 32);
            const unsigned xoff = (unsigned)m * (unsigned)g.D * 4u + hoff;   // (M D 4 < 2^32: vfm_vit_forward checks)
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) e.x[grp] = *reinterpret_cast<const float4*>(xb + (xoff + 32u * grp));
        }
    }
    if constexpr (EPI == EPI_PATCH) {
        if (t < g.T) {
            const char* cb = reinterpret_cast<const char*>(g.clspos + n32u * 32);
            const unsigned coff = (unsigned)t * (unsigned)g.D * 4u + hoff;
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) e.x[grp] = *reinterpret_cast<const float4*>(cb + (coff + 32u * grp));   // row 0 = cls + pos[0]
        }
    }
}
// the token's LayerNorm statistics from the producers' partial sums, slices in ascending order: (a, nb) = (rstd, -rstd mean), so that
// LN(x) W^T = a (x W'^T) + nb c + b'
__device__ __forceinline__ void ln_stats_load(const GemmArgs& g, int m, float& ln_a, float& ln_nb) {
    const int nsl = g.D / 32;
    const float4* sp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(g.stats) + (unsigned)m * (unsigned)nsl * 8u);   // two slices per float4
    float sx = 0.f, sq = 0.f;
    if (nsl <= 12) {   // ViT-S: six float4 instead of sixteen predicated ones and their 64 additions (the same sums: the rest were + 0)
        float4 st[6];
        if (nsl == 12) {   // (the usual case without a branch per load)
#pragma unroll
            for (int i = 0; i < 6; ++i) st[i] = sp[i];
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) st[i] = (2 * i < nsl) ? sp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            sx += st[i].x; sq += st[i].y;
            sx += st[i].z; sq += st[i].w;
        }
    } else {
        float4 st[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = (2 * i < nsl) ? sp[i] : make_float4(0.f, 0.f, 0.f, 0.f);   // (D <= 1024: 32 slices)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sx += st[i].x; sq += st[i].y;
            sx += st[i].z; sq += st[i].w;
        }
    }
    const float mean = sx * g.invD;
    const float var = fmaxf(__builtin_fmaf(sq, g.invD, -(mean * mean)), 0.0f);
    ln_a = __builtin_amdgcn_rsqf(var + 1e-6f);   // (the argument is >= 1e-6: no denormal guard)
    ln_nb = -(ln_a * mean);
}
// NT: the fp16 outputs leave with non-temporal stores (vfm_debug_set_vit_gemm(-13, 1), token-stationary kernel only: A/B)
__device__ __forceinline__ void store_half4(void* p, half4 v, bool nt) {
    if (nt) {
        const uint2 u = *reinterpret_cast<const uint2*>(&v);
        __builtin_nontemporal_store(u.x, reinterpret_cast<unsigned*>(p));
        __builtin_nontemporal_store(u.y, reinterpret_cast<unsigned*>(p) + 1);
    } else {
        *reinterpret_cast<half4*>(p) = v;
    }
}
// mt, b, tq, n32: wave-uniform (the callers pass them through readfirstlane); (ln_a, ln_nb) of the lane's token
template <int EPI, bool NT = false>
__device__ __forceinline__ void epi_tile(const GemmArgs& g, const floatx16& acc, int m, int mt, int b, int tq, int t, int hi, int n32,
                                         const EpiRegs& e, float ln_a, float ln_nb) {
    const int lane31 = lane_id() & 31;
    // bytes: row lane31 of a 32 x 8 fragment row (16 B), halves 4 hi ..; a fragment-tiled [M][K] matrix with ks = K / 16 k-steps keeps
    // element (m, n0 .. n0 + 3) of this lane and group at ((mt ks 2 + n32 4 + grp) 256 halves + lane part   (= frag_index(m, n0, ks))
    const unsigned lane_off = (unsigned)lane31 * 16u + 8u * (unsigned)hi;
    const f2 A2 = splat2(ln_a), NB2 = splat2(ln_nb);
    if constexpr (EPI == EPI_QKV) {
        const int hg = n32 >> 1;                                  // 64-channel unit = (which, head)
        const int which = hg >= 2 * g.heads ? 2 : (hg >= g.heads ? 1 : 0);
        const int head = hg - which * g.heads;
        // (offsets in 32 bits -- vfm_vit_forward checks that every activation buffer is below 4 GiB --, one 64-bit add per tile)
        const unsigned bh_off = ((unsigned)b * (unsigned)g.heads + (unsigned)head) * (unsigned)g.Tp * 64u;   // halves; q, k: [b heads + head][Tp][64], 4 k-steps
        if (which < 2) {   // (tq 4 + s) 2 + h with s = (n32 & 1) 2 + (grp >> 1), h = grp & 1
            char* base = reinterpret_cast<char*>(which == 0 ? g.q : g.k) + (size_t)((bh_off + ((unsigned)tq * 8u + (unsigned)(n32 & 1) * 4u) * 256u) * 2u);
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const f2 v01 = fma2(f2{acc[grp * 4 + 0], acc[grp * 4 + 1]}, A2, fma2(NB2, f2{e.aux[grp].x, e.aux[grp].y}, f2{e.bias[grp].x, e.bias[grp].y}));
                const f2 v23 = fma2(f2{acc[grp * 4 + 2], acc[grp * 4 + 3]}, A2, fma2(NB2, f2{e.aux[grp].z, e.aux[grp].w}, f2{e.bias[grp].z, e.bias[grp].w}));
                store_half4(base + (lane_off + 512u * grp), to_half4(v01, v23), NT);
            }
        } else {
            // V^T [b heads + head][64][Tp], Tp / 16 k-steps: row d = (n32 & 1) 32 + 8 grp + 4 hi + j, key t: tile n32 & 1, k-step
            // 2 tq + (lane31 >> 4), half (lane31 >> 3) & 1, element lane31 & 7
            char* base = reinterpret_cast<char*>(g.vt) + (size_t)((bh_off + ((unsigned)(n32 & 1) * (unsigned)(g.Tp / 16) * 2u + (unsigned)tq * 4u) * 256u) * 2u);
            const unsigned voff = ((unsigned)(lane31 >> 3) * 256u + (unsigned)(4 * hi) * 8u + (unsigned)(lane31 & 7)) * 2u;
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const f2 v01 = fma2(f2{acc[grp * 4 + 0], acc[grp * 4 + 1]}, A2, fma2(NB2, f2{e.aux[grp].x, e.aux[grp].y}, f2{e.bias[grp].x, e.bias[grp].y}));
                const f2 v23 = fma2(f2{acc[grp * 4 + 2], acc[grp * 4 + 3]}, A2, fma2(NB2, f2{e.aux[grp].z, e.aux[grp].w}, f2{e.bias[grp].z, e.bias[grp].w}));
                const half4 o = to_half4(v01, v23);
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<_Float16*>(base + (voff + (unsigned)(8 * grp + j) * 16u)) = o[j];
            }
        }
    } else if constexpr (EPI == EPI_GELU) {
        char* base = reinterpret_cast<char*>(g.out) + (size_t)((((unsigned)mt * (unsigned)(g.N / 16) * 2u + (unsigned)n32 * 4u) * 256u) * 2u);
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const f2 v01 = fma2(f2{acc[grp * 4 + 0], acc[grp * 4 + 1]}, A2, fma2(NB2, f2{e.aux[grp].x, e.aux[grp].y}, f2{e.bias[grp].x, e.bias[grp].y}));
            const f2 v23 = fma2(f2{acc[grp * 4 + 2], acc[grp * 4 + 3]}, A2, fma2(NB2, f2{e.aux[grp].z, e.aux[grp].w}, f2{e.bias[grp].z, e.bias[grp].w}));
            store_half4(base + (lane_off + 512u * grp), to_half4(gelu2(v01), gelu2(v23)), NT);
        }
    } else {   // PRODUCES_LN: the residual stream's new values, their fp16 copy, the slice's (sum x, sum x^2)
        char* fbase = reinterpret_cast<char*>(g.xh) + (size_t)((((unsigned)mt * (unsigned)(g.D / 16) * 2u + (unsigned)n32 * 4u) * 256u) * 2u);
        char* xb = reinterpret_cast<char*>(g.x + n32 * 32);
        const unsigned xoff = (unsigned)m * (unsigned)g.D * 4u + 16u * (unsigned)hi;
        f2 ps = splat2(0.f), pq = splat2(0.f);   // this lane's share of the slice's (sum x, sum x^2), pairwise
        const bool row = t < g.T;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            f2 o01 = f2{e.x[grp].x, e.x[grp].y}, o23 = f2{e.x[grp].z, e.x[grp].w};   // (padding rows stay exactly zero)
            const f2 v01 = f2{acc[grp * 4 + 0], acc[grp * 4 + 1]} + f2{e.bias[grp].x, e.bias[grp].y};
            const f2 v23 = f2{acc[grp * 4 + 2], acc[grp * 4 + 3]} + f2{e.bias[grp].z, e.bias[grp].w};
            if constexpr (EPI == EPI_PATCH) {
                if (t != 0 && row) {   // (t == 0: cls + pos[0] as loaded)
                    o01 = v01 + o01;
                    o23 = v23 + o23;
                }
            } else if (row) {
                o01 = fma2(f2{e.aux[grp].x, e.aux[grp].y}, v01, o01);
                o23 = fma2(f2{e.aux[grp].z, e.aux[grp].w}, v23, o23);
            }
            if ((EPI == EPI_PATCH || row) && !(g.hot_a & 4)) *reinterpret_cast<float4*>(xb + (xoff + 32u * grp)) = make_float4(o01[0], o01[1], o23[0], o23[1]);
            if (!(g.hot_a & 8)) *reinterpret_cast<half4*>(fbase + (lane_off + 512u * grp)) = to_half4(o01, o23);
            ps += o01 + o23;
            pq = fma2(o01, o01, fma2(o23, o23, pq));
        }
        // the two half-waves hold the slice's other 16 channels of the same token: lower + upper, in that order
        float psum = ps[0] + ps[1], psq = pq[0] + pq[1];
        const float osum = __shfl_xor(psum, 32), osq = __shfl_xor(psq, 32);
        if (hi == 0) *reinterpret_cast<float2*>(reinterpret_cast<char*>(g.stats) + ((unsigned)m * (unsigned)(g.D / 32) + (unsigned)n32) * 8u) = make_float2(psum + osum, psq + osq);
    }
}

// ---- helpers of the LDS-staged kernels
template <int N>
__device__ __forceinline__ void vit_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void vit_glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst_uniform)
        : "memory");
}
__device__ __forceinline__ unsigned pack_f16x2(float a, float b) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    half2v h;
    h[0] = (_Float16)a;
    h[1] = (_Float16)b;
    return *reinterpret_cast<unsigned*>(&h);
}


// This file is compiled TWICE (build.py): VFM_VIT_PART 0 -- everything but the fused MLP kernel, with -mllvm -amdgpu-mfma-vgpr-form (MFMA results
// in architectural registers: vit_qkv_attention_kernel) -- and VFM_VIT_PART 1 -- vit_mlp_kernel alone, WITHOUT that option: its 192 output
// accumulators live in the accumulation file beside 256 architectural registers of everything else.  The two objects share the code above
// (internal linkage in each) and meet at vfm_vit_launch_mlp_.
#ifndef VFM_VIT_PART
#define VFM_VIT_PART 0
#endif
}  // namespace
extern "C" __attribute__((visibility("hidden"))) int vfm_vit_launch_mlp_(const void* fc1_args, const void* fc2_args, void* stream);
namespace {
#if VFM_VIT_PART == 0

// NT = 32-channel tiles per wave (2 for the wide GEMMs, 1 for N = dim so that 66 x 12 = 792 waves
// cover the chip instead of 396).
template <int EPI, int NT, int PF>
__global__ __launch_bounds__(256) void vit_gemm_kernel(GemmArgs g) {
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int ntiles = g.N / (32 * NT);
    const int wid = blockIdx.x * (int)(blockDim.x >> 6) + wave;   // (1, 2 or 4 waves per workgroup: launch_gemm_cfg)
    int mt = wid / ntiles, nt = wid % ntiles;  // token tile, (32*NT)-channel tile
    if (g.xcd_map) {
        if (!xcd_item(g.M / 32, ntiles, wave, mt, nt)) return;
    } else if (mt >= g.M / 32) return;
    const uint4* Ap = g.A + (size_t)mt * g.KS * 64 + lane;
    const uint4* Wp = g.W + (size_t)(nt * NT) * g.KS * 64 + lane;
    floatx16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // The grid is ~1 wave per SIMD, so nothing hides a load's L2 round trip except the wave itself:
    // keep PF k-steps of operand fragments in flight in a register ring.
    // Round 5: every load of the ring is UNCONDITIONAL (k-steps past the end re-read the last one) and stays where it is issued
    // (sched_barrier), so that the compiler can count the loads in flight.  With the ring's loads under `if (s + PF < KS)` it waited for
    // vmcnt(0) at the head of every block of PF k-steps: each block began by exposing the L2 round trip of the fragments it had just
    // asked for (and the token-stationary kernels did so in front of every k-step).
    uint4 ra[PF], rw[NT][PF];
    const int klast = g.KS - 1;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        const int si = i < klast ? i : klast;
        ra[i] = Ap[(size_t)si * 64];
#pragma unroll
        for (int j = 0; j < NT; ++j) rw[j][i] = Wp[((size_t)j * g.KS + si) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int m = mt * 32 + (lane & 31);
    const int mtu = __builtin_amdgcn_readfirstlane(mt), qtiles = g.Tp >> 5;   // (a token tile lies inside one image: Tp % 32 == 0)
    const int b = (int)(((unsigned)mtu * g.qt_magic) >> 20), tq = mtu - b * qtiles, t = tq * 32 + (lane & 31);
    const int hi = lane >> 5;
    // Everything the epilogue reads is requested HERE, behind the first operand fragments and in front of the k-loop (round 4): as
    // the epilogue's own loads -- bias, row sums, LayerScale, the residual values, twelve partial sums in a loop the compiler could
    // not unroll -- they were one more chain of L2 round trips at the end of every kernel of a forward that is nothing but such chains.
    EpiRegs e[NT];
#pragma unroll
    for (int half = 0; half < NT; ++half) epi_load<EPI>(g, m, t, hi, __builtin_amdgcn_readfirstlane(nt * NT + half), e[half]);
    float ln_mean = 0.f, ln_rstd = 0.f;   // (a, nb) of the lane's token: see ln_stats_load
    if constexpr (EPI == EPI_QKV || EPI == EPI_GELU) ln_stats_load(g, m, ln_mean, ln_rstd);
    const int nblk = (g.KS + PF - 1) / PF;
    for (int blk = 0; blk + 1 < nblk; ++blk) {   // every block but the last: multiply, reload the slot for the next block
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int s = blk * PF + i;
            const half8 av = *reinterpret_cast<half8*>(&ra[i]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&rw[j][i]), av, acc[j], 0, 0, 0);
            const int sn = s + PF < klast ? s + PF : klast;
            ra[i] = Ap[(size_t)sn * 64];
#pragma unroll
            for (int j = 0; j < NT; ++j) rw[j][i] = Wp[((size_t)j * g.KS + sn) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // the last block: no reloads (nothing is in flight when the epilogue starts); a partial one where KS % PF != 0 (the patch embedding's 37)
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        if ((nblk - 1) * PF + i < g.KS) {
            const half8 av = *reinterpret_cast<half8*>(&ra[i]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&rw[j][i]), av, acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int half = 0; half < NT; ++half)
        epi_tile<EPI>(g, acc[half], m, mtu, b, tq, t, hi, __builtin_amdgcn_readfirstlane(nt * NT + half), e[half], ln_mean, ln_rstd);
}

// ---- the same product with workgroup tiles of 128 tokens x 128 channels staged through the LDS (round 4): for
// batches of many images.  The direct kernel above moves one A and one W fragment (2 KiB) from the L2 per MFMA -- at 96 images the
// forward sits at 290 TFLOP/s on ~9 TB/s of L2 traffic --; here a k-step's eight fragments (4 token tiles + 4 channel tiles, 8 KiB)
// feed the 16 MFMAs of four waves: 0.5 KiB per MFMA.  Four waves, each 64 tokens x 64 channels (2 x 2 MFMA tiles); stages of KB = 4
// k-steps (32 KiB: 32 LDS-DMA pieces of one (tile, k-step) fragment row each), two stages in the LDS, one barrier per stage: the
// pieces of stage i + 1 are issued behind the barrier that ends stage i - 1's reads and land under stage i's 16 MFMAs per wave.
// At one scan (6 images: 17 x 3 workgroups for N = 384) the direct kernel's spread wins; launch_gemm picks by workgroup count.
// KB k-steps per stage, NS stages in the LDS ring (NS - 1 in flight or in use beside the one being filled): a stage is issued NS - 2
// stages of MFMAs before it is needed -- an L2 round trip is ~1500 cycles, a stage's MFMAs 128 KB cycles per wave -- and 4 KB x NS KiB
// of LDS lets several workgroups share a compute unit (KB = 2, NS = 4: 64 KiB, 2 per CU; KB = 2, NS = 3: 48 KiB, 3 per CU).
// (__launch_bounds__(256, 3): without the bound the compiler kept the 64 accumulators in AGPRs beside 116 - 140 VGPRs and moved them back and
// forth -- 180 - 204 registers, two waves per SIMD; with it 112 - 137 and no AGPR, no spill: three waves per SIMD with a ring of three stages
// (48 KiB): 96 images 4.20 -> 4.09 ms, 48 images 2.39 -> 2.29)
// NG (round 5): channel groups of 128 a workgroup takes -- 1: a 128 x 128 tile (three workgroups per compute unit); 3: all 384 channels of
// the residual GEMMs (proj, fc2) for its 128 tokens: one workgroup per compute unit, 64 x 192 per wave.  Why: LDS-DMA moves at most
// ~64 bytes per clock and compute unit (tools/probe/l2_lds_probe.hip: 125 - 145 GB/s per unit from the L2 whatever is in flight), and a
// 128 x 128 tile asks for exactly that at the matrix pipes' peak -- (128 + 128) x 32 B per k-step of 4 x 4 MFMAs = 64 B per clock:
// the loop can be at best half address path, half MFMA (measured: a stage's waves spend half their time between the barrier and
// their fragments' arrival, tools/trace_vit_lds.py).  128 x 384 asks for 43 B per clock, reads the token operand once instead of
// three times, and 237 equal workgroups on 256 units have no tail (with three 128 x 128 workgroups per unit the last one ended 17 us
// behind the median of fc2's 62).
// NW waves per workgroup (4, or 8 for the wide tile: 64 x 96 per wave, two waves per SIMD -- with four waves of 64 x 192 a SIMD's one
// wave did its stage's LDS-DMA issue + fragment reads and its 24 MFMAs one after the other: 661 + 820 cycles per stage, slower than
// the 128 x 128 form)
// DBG: the tools' trace (s_memtime in the loop makes the compiler drain the LDS queue at every wait: an instantiation of its own)
template <int EPI, int KB, int NS, int NG = 1, int NW = 4, bool DBG = false>
__global__ __launch_bounds__(64 * NW, NG == 1 ? 3 : 1) void vit_gemm_lds_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [stage][A | W][tile][k-step][1 KiB]
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mtiles = g.M / 32, ngroups = g.N / (128 * NG), mgroups = (mtiles + 3) / 4;
    // group of four token tiles mg on XCD mg % 8 (workgroup b runs on XCD b % 8: observed, speed only): the workgroups that share
    // its A tiles -- one per channel group -- read them through one L2
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mg = xcd + 8 * (idx / ngroups), ng = idx % ngroups;
    if (mg >= mgroups) return;
    unsigned long long dbg_t0 = 0, dbg_t1 = 0;
    if (DBG && g.dbg) dbg_t0 = wall_clock64();
    const int wm = wave / (NW / 2), wn = wave % (NW / 2);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    constexpr int STAGE = (1 + NG) * 4 * KB * 1024;
    constexpr int PW = (4 + 4 * NG) * KB / NW;   // pieces per wave and stage: (4 + 4 NG) KB fragment rows over NW waves
    const int nstages = (g.KS + KB - 1) / KB;
    // piece p of a stage: operand p / (4 KB) (0 = A, 1 = W), tile (p / KB) % 4, k-step p % KB; wave w issues pieces w, w + 4, ...
    // Round 5 (the loop spent 45 scalar instructions per stage of 8 MFMAs on these addresses): a piece's source is a wave-uniform
    // 64-bit pointer kept in scalar registers -- set up once -- plus a 32-bit offset (the saddr form of global_load_lds): the lane's 16
    // bytes + KB KiB per stage issued so far; the pieces of a stage go out in one block that saves and restores m0 once.
    // A k-step past the end keeps its pointer (re-reads a valid fragment row; the piece count per stage is a constant, which is what
    // the counted waits below rest on; nothing reads such bytes).
    const char* src[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int p = wave + NW * i;
        const int op = p < 4 * KB ? 0 : 1, tile = op == 0 ? p / KB : (p - 4 * KB) / KB;   // (A: four tiles; W: 4 NG tiles)
        int ks = p % KB;
        ks = ks < g.KS ? ks : g.KS - 1;
        int rowtile = op == 0 ? mg * 4 + tile : ng * 4 * NG + tile;
        if (op == 0 && rowtile >= mtiles) rowtile = mtiles - 1;   // (a partial last group of token tiles: its epilogue is skipped)
        if (op == 0 && (g.hot_a & 1)) rowtile = tile;                   // (timing experiment: tools/ab_vit_hot_a.sh)
        src[i] = reinterpret_cast<const char*>((op == 0 ? g.A : g.W) + ((size_t)rowtile * g.KS + ks) * 64);
    }
    unsigned lane16 = (unsigned)lane * 16u;   // + the bytes the wave's pieces have moved on by (one VALU add per stage instead of PW 64-bit scalar ones)
    const int kk_wave = wave % KB;
    int ks_next = 0;   // first k-step of the stage issue_stage sends next
    // a stage goes out piece by piece, BETWEEN the MFMAs of the stage in use (issue_piece): issued in one block right behind the
    // barrier the PW LDS-DMA instructions of every wave filled the vector-memory queue at once -- the address path moves ~64 bytes per
    // clock and compute unit, a stage is 16 - 32 KiB: 250 - 500 cycles -- and the fragment reads queued up behind them: a stage's waves
    // spent as long between the barrier and their fragments' arrival as in their MFMAs (tools/trace_vit_lds.py)
    unsigned dst_cur = 0;
    auto begin_stage = [&](int slot) { dst_cur = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(slot * STAGE) + (unsigned)wave * 1024u); };
    auto issue_piece = [&](int i) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(src[i]), "s"(dst_cur + (unsigned)(NW * 1024) * (unsigned)i) : "memory");
    };
    auto end_stage = [&]() {
        ks_next += KB;
        // (wave + NW i) % KB = wave % KB: every piece of this wave is the same k-step of its stage; it moves on while its next k-step exists
        lane16 += (ks_next + kk_wave < g.KS) ? (unsigned)(KB * 1024) : 0u;
    };
    auto issue_stage = [&](int slot) {
        begin_stage(slot);
#pragma unroll
        for (int i = 0; i < PW; ++i) issue_piece(i);
        end_stage();
    };
    constexpr int NJ = 8 * NG / NW;   // channel tiles per wave
    floatx16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int st = 0; st < NS; ++st) issue_stage(st);   // stages 0 .. NS - 1: every slot of the ring (dummies past the end)
    // Fragments double-buffered in registers across the stage barrier (round 5).  A stage's fragments are all read into registers
    // before its MFMAs; so behind the barrier that says "stage s + 1 has landed" everybody has READ stage s, its slot takes stage
    // s + NS at once, and the reads of stage s + 1 go out in front of the MFMAs of stage s: LDS reads (80 KiB per stage and compute
    // unit for the wide tile: 310 cycles at the LDS's 256 B per clock), LDS-DMA (16 - 32 KiB at ~64 B per clock) and MFMAs (768 cycles)
    // run under one another instead of one after the other -- the barrier kept the waves of a workgroup in the same phase, and the
    // phases added up (tools/trace_vit_lds.py: 26 + 114 + 554 + 543 cycles per stage).
    struct Frags {
        half8 a[KB][2], w[KB][NJ];
    };
    auto read_frags = [&](Frags& f, int sl) {
        const unsigned char* st = lds + sl * STAGE;
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) f.a[ks][i] = *reinterpret_cast<const half8*>(st + ((2 * wm + i) * KB + ks) * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < NJ; ++j) f.w[ks][j] = *reinterpret_cast<const half8*>(st + 4 * KB * 1024 + ((NJ * wn + j) * KB + ks) * 1024 + lane * 16);
        }
    };
    int slot = 0;   // slot of the stage whose fragments are in registers
    unsigned long long c_wait = 0, c_bar = 0, c_mfma = 0;   // (DBG: counter ticks of the last wave at the DMA wait, the LDS wait + barrier, everything else)
    auto body = [&](const Frags& cur, Frags& nxt, int stage) {
        unsigned long long tc0 = 0, tc1 = 0, tc2 = 0;
        const bool more = stage + 1 < nstages;   // wave-uniform, the same in every wave
        if (DBG && g.dbg) tc0 = __builtin_readcyclecounter();
        if (more) {
            vit_wait_vmcnt<(NS - 2) * PW>();      // this wave's pieces of stage + 1: the NS - 2 stages issued since may fly on
            if (DBG && g.dbg) tc1 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of stage `stage` (issued a stage of MFMAs ago) have returned
            __builtin_amdgcn_s_barrier();         // everybody's; and everybody holds stage `stage` in registers: its slot is free
            asm volatile("" ::: "memory");
            if (DBG && g.dbg) tc2 = __builtin_readcyclecounter();
            begin_stage(slot);                    // stage + NS
            read_frags(nxt, slot + 1 == NS ? 0 : slot + 1);
        }
        constexpr int NMFMA = KB * 2 * NJ, EVERY = NMFMA / PW > 0 ? NMFMA / PW : 1;
        int piece = 0;
#pragma unroll
        for (int ks = 0; ks < KB; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.w[ks][j], cur.a[ks][i], acc[i][j], 0, 0, 0);
                    const int q = (ks * 2 + i) * NJ + j;
                    if (q % EVERY == 0 && piece < PW) {   // (compile-time: the loops are unrolled)
                        if (more) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue_piece(piece);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        ++piece;
                    }
                }
        if (more) {
#pragma unroll
            for (int i2 = 0; i2 < PW; ++i2)
                if (i2 >= piece) issue_piece(i2);
            end_stage();
        }
        if (DBG && g.dbg && more) {
            const unsigned long long tc3 = __builtin_readcyclecounter();
            c_wait += tc1 - tc0;
            c_bar += tc2 - tc1;
            c_mfma += tc3 - tc2;
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
    };
    Frags f0, f1;
    vit_wait_vmcnt<(NS - 1) * PW>();   // stage 0
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags(f0, 0);
    int stage = 0;
    for (; stage + 1 < nstages; stage += 2) {
        body(f0, f1, stage);
        body(f1, f0, stage + 1);
    }
    if (stage < nstages) body(f0, f1, stage);
    vit_wait_vmcnt<0>();   // (the dummy stages: no LDS-DMA may be in flight when the workgroup's LDS is handed on)
    if (DBG && g.dbg) dbg_t1 = wall_clock64();
    const int hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mt = mg * 4 + 2 * wm + i;
        if (mt >= mtiles) continue;   // wave-uniform
        const int m = mt * 32 + (lane & 31);
        const int mtu = __builtin_amdgcn_readfirstlane(mt), qtiles = g.Tp >> 5;
        const int b = (int)(((unsigned)mtu * g.qt_magic) >> 20), tq = mtu - b * qtiles, t = tq * 32 + (lane & 31);
        float ln_mean = 0.f, ln_rstd = 0.f;   // (a, nb) of the lane's token: see ln_stats_load
        if constexpr (EPI == EPI_QKV || EPI == EPI_GELU) ln_stats_load(g, m, ln_mean, ln_rstd);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n32 = __builtin_amdgcn_readfirstlane(ng * 4 * NG + NJ * wn + j);
            EpiRegs e;
            epi_load<EPI>(g, m, t, hi, n32, e);
            epi_tile<EPI>(g, acc[i][j], m, mtu, b, tq, t, hi, n32, e, ln_mean, ln_rstd);
        }
    }
    if (DBG && g.dbg && wave == NW - 1 && lane == 0) {   // (tools/trace_vit_lds.py) start, end of the k loop, end: 100 MHz ticks
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g.dbg[blockIdx.x * 4 + 0] = dbg_t0;
        g.dbg[blockIdx.x * 4 + 1] = dbg_t1;
        g.dbg[blockIdx.x * 4 + 2] = wall_clock64();
        g.dbg[blockIdx.x * 4 + 3] = (unsigned long long)g.KS;
        unsigned long long* more = g.dbg + 4 * 4096 + blockIdx.x * 4;   // (the tools' buffer holds [2][4096][4])
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        more[0] = c_wait;
        more[1] = c_bar;
        more[2] = ((unsigned long long)(xcc & 0xf) << 32) | hw;   // where the workgroup ran
        more[3] = c_mfma;
    }
}

// ---- the wide K = dim products (QKV, fc1) of LARGE batches with the token rows stationary (end of round 4).  The 128 x 128 kernel above
// fetches every A tile N / 128 times and every W tile M / 128 times from the L2: 0.5 KiB per MFMA, 370 - 500 MB per GEMM at 96 images, and
// it runs at the ~5.5 TB/s the L2 -> LDS path delivers to it (fc2, with four times the k-loop per tile, reaches 7.8).  Here a workgroup keeps
// its 128 token rows -- 4 tiles x KS fragment rows = 96 KiB at K = 384 -- in the LDS for the whole kernel and its NW waves share nothing
// else: wave w takes channel tiles w, w + NW, ... , streams THEIR weight fragments through a register ring (plain loads: the compiler
// counts them) and multiplies each with the four A fragments of the k-step from the LDS.  0.25 KiB per MFMA from the L2, no barrier after
// the first, a wave's epilogue stalls nobody else.  One workgroup per compute unit: it pays when its rounds of 256 workgroups are full
// (launch_gemm's policy; measured per batch size in tools/ab_vit_astat.py).
// Same MFMAs over the same fragments in the same k order: bit-identical to the other two kernels.
template <int EPI, int NW, int PF, bool NTS = false>
__global__ __launch_bounds__(64 * NW, 1) void vit_gemm_astat_kernel(GemmArgs g) {
    static_assert(EPI == EPI_QKV || EPI == EPI_GELU, "the residual epilogues read a row-sized operand per tile: not this kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [4 token tiles][KS][1 KiB]
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mtiles = g.M / 32, KS = g.KS, mg = blockIdx.x;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    unsigned long long dbg_t0 = 0;
    if (g.dbg) dbg_t0 = wall_clock64();
    for (int p = wave; p < 4 * KS; p += NW) {
        const int tile = p / KS, ks = p - tile * KS;
        int rowtile = mg * 4 + tile;
        if (rowtile >= mtiles) rowtile = mtiles - 1;   // (a partial last group: its epilogue is skipped)
        vit_glds16(g.A + ((size_t)rowtile * KS + ks) * 64 + lane, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)p * 1024u));
    }
    const int hi = lane >> 5, lane31 = lane & 31;
    float ln_mean[4], ln_rstd[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int mt = mg * 4 + i;
        if (mt >= mtiles) mt = mtiles - 1;
        ln_stats_load(g, mt * 32 + lane31, ln_mean[i], ln_rstd[i]);
    }
    vit_wait_vmcnt<0>();
    __syncthreads();
    const int ntiles = g.N / 32, qtiles = g.Tp >> 5;
    for (int nt = wave; nt < ntiles; nt += NW) {
        const uint4* Wp = g.W + (size_t)nt * KS * 64 + lane;
        uint4 rw[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) rw[i] = Wp[(size_t)(i < KS - 1 ? i : KS - 1) * 64];   // (unconditional: see vit_gemm_kernel)
        __builtin_amdgcn_sched_barrier(0);
        const int n32 = __builtin_amdgcn_readfirstlane(nt);
        EpiRegs e;
        epi_load<EPI>(g, 0, 0, hi, n32, e);   // bias + row sums of the folded weight: per channel, the same for the four token tiles
        floatx16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int nblk = (KS + PF - 1) / PF;
        for (int blk = 0; blk + 1 < nblk; ++blk) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int s = blk * PF + i;
                const half8 wv = *reinterpret_cast<half8*>(&rw[i]);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const half8 af = *reinterpret_cast<const half8*>(lds + ((size_t)(t4 * KS + s)) * 1024 + lane * 16);
                    acc[t4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv, af, acc[t4], 0, 0, 0);
                }
                rw[i] = Wp[(size_t)(s + PF < KS - 1 ? s + PF : KS - 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) {   // the last block (partial where KS % PF != 0): no reloads
            const int s = (nblk - 1) * PF + i;
            if (s < KS) {
                const half8 wv = *reinterpret_cast<half8*>(&rw[i]);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const half8 af = *reinterpret_cast<const half8*>(lds + ((size_t)(t4 * KS + s)) * 1024 + lane * 16);
                    acc[t4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv, af, acc[t4], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int mt = mg * 4 + t4;
            if (mt >= mtiles) continue;   // workgroup-uniform
            const int mtu = __builtin_amdgcn_readfirstlane(mt);
            const int b = (int)(((unsigned)mtu * g.qt_magic) >> 20), tq = mtu - b * qtiles;
            epi_tile<EPI, NTS>(g, acc[t4], mt * 32 + lane31, mtu, b, tq, tq * 32 + lane31, hi, n32, e, ln_mean[t4], ln_rstd[t4]);
        }
    }
    if (g.dbg && wave == NW - 1 && lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g.dbg[blockIdx.x * 4 + 0] = dbg_t0;
        g.dbg[blockIdx.x * 4 + 1] = wall_clock64();
        g.dbg[blockIdx.x * 4 + 2] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
        g.dbg[blockIdx.x * 4 + 3] = 0;
    }
}

// ---- the token-stationary product with TWO channel tiles per wave (round 5).  What the counters said once the epilogues were on a diet
// (profiles/r05_pmc_vit_96images.json): the GEMMs issue few instructions and still keep the matrix pipe 22 - 24 % busy -- every MFMA of the
// kernel above is fed by one 1 KiB ds_read_b128 of an A fragment, i.e. 32 cycles of the LDS port (128 B / clk per compute unit) per 32-cycle
// MFMA on each of four SIMDs: the LDS runs at its peak exactly when the matrix pipes do, and neither gets there.  Here a wave holds the weight
// fragments of two channel tiles and multiplies each A fragment with both: half the LDS bytes per MFMA (8 MFMAs per four ds_reads), 128
// accumulator registers, two waves per SIMD (eight per workgroup).  A wave takes the channel-tile pairs w, w + 8, ... (36 tiles of QKV = 18
// pairs: waves 0 and 1 run a third round).  The epilogue's per-channel operands are fetched behind the k-loop, when the weight ring's
// registers are free.  Same MFMAs over the same fragments in the same k order: bit-identical to the other GEMM kernels.
template <int EPI, int NW, int PF>
__global__ __launch_bounds__(64 * NW, 1) void vit_gemm_astat2_kernel(GemmArgs g) {
    static_assert(EPI == EPI_QKV || EPI == EPI_GELU, "the residual epilogues read a row-sized operand per tile: not this kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [4 token tiles][KS][1 KiB]
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mtiles = g.M / 32, KS = g.KS, mg = blockIdx.x;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    for (int p = wave; p < 4 * KS; p += NW) {
        const int tile = p / KS, ks = p - tile * KS;
        int rowtile = mg * 4 + tile;
        if (rowtile >= mtiles) rowtile = mtiles - 1;   // (a partial last group: its epilogue is skipped)
        vit_glds16(g.A + ((size_t)rowtile * KS + ks) * 64 + lane, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)p * 1024u));
    }
    const int hi = lane >> 5, lane31 = lane & 31;
    float ln_a[4], ln_nb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int mt = mg * 4 + i;
        if (mt >= mtiles) mt = mtiles - 1;
        ln_stats_load(g, mt * 32 + lane31, ln_a[i], ln_nb[i]);
    }
    vit_wait_vmcnt<0>();
    __syncthreads();
    const int npairs = g.N / 64, qtiles = g.Tp >> 5;
    const unsigned char* la = lds + lane * 16;
    for (int pr = wave; pr < npairs; pr += NW) {
        const uint4* W0 = g.W + (size_t)(2 * pr) * KS * 64 + lane;
        const uint4* W1 = W0 + (size_t)KS * 64;
        uint4 rw0[PF], rw1[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {   // (KS >= PF: unconditional, in k-step order -- the order the loop's counted waits assume)
            rw0[i] = W0[(size_t)i * 64];
            rw1[i] = W1[(size_t)i * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        floatx16 acc0[4], acc1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[i][r] = acc1[i][r] = 0.f;
        // (KS % PF == 0 -- launch_gemm checks -- and the ring's reloads are unconditional, the last PF of them re-reading the final k-step:
        // a straight-line loop body, so that the compiler can COUNT the loads in flight.  With the reload under `if (s + PF < KS)` it waited
        // for vmcnt(0) in front of every k-step -- every weight fragment's L2 round trip was exposed, 24 times per tile pair: that, not
        // the LDS and not the issue rate, was what the token-stationary kernels' 50 - 66 us consisted of.)
        // The token fragments of k-step s + 1 are requested in front of the MFMAs of k-step s (two register sets, alternating: PF is even
        // and every block starts at a multiple of PF): re-used registers had every k-step begin with its own four LDS reads and a wait
        // for the first of them -- ~150 cycles in front of 256 cycles of MFMAs, with one other wave per SIMD to fill them.
        static_assert(PF % 2 == 0, "two alternating fragment sets");
        half8 af[2][4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) af[0][t4] = *reinterpret_cast<const half8*>(la + ((size_t)(t4 * KS)) * 1024);
        for (int s0 = 0; s0 + PF < KS; s0 += PF) {   // every block of PF k-steps but the last
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int s = s0 + i;
                const half8 w0 = *reinterpret_cast<half8*>(&rw0[i]);
                const half8 w1 = *reinterpret_cast<half8*>(&rw1[i]);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) af[(i + 1) & 1][t4] = *reinterpret_cast<const half8*>(la + ((size_t)(t4 * KS + s + 1)) * 1024);
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    acc0[t4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, af[i & 1][t4], acc0[t4], 0, 0, 0);
                    acc1[t4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, af[i & 1][t4], acc1[t4], 0, 0, 0);
                }
                rw0[i] = W0[(size_t)(s + PF) * 64];
                rw1[i] = W1[(size_t)(s + PF) * 64];
                __builtin_amdgcn_sched_barrier(0);   // (the reload stays HERE, behind its k-step: the scheduler had gathered all PF pairs at the loop's end)
            }
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) {   // the last block: nothing left to fetch
            const int s = KS - PF + i;
            const half8 w0 = *reinterpret_cast<half8*>(&rw0[i]);
            const half8 w1 = *reinterpret_cast<half8*>(&rw1[i]);
            if (i + 1 < PF) {
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) af[(i + 1) & 1][t4] = *reinterpret_cast<const half8*>(la + ((size_t)(t4 * KS + s + 1)) * 1024);
            }
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                acc0[t4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, af[i & 1][t4], acc0[t4], 0, 0, 0);
                acc1[t4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, af[i & 1][t4], acc1[t4], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n32 = __builtin_amdgcn_readfirstlane(2 * pr + j);
            EpiRegs e;
            epi_load<EPI>(g, 0, 0, hi, n32, e);   // bias + row sums of the folded weight: per channel, the same for the four token tiles
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const int mt = mg * 4 + t4;
                if (mt >= mtiles) continue;   // workgroup-uniform
                const int mtu = __builtin_amdgcn_readfirstlane(mt);
                const int b = (int)(((unsigned)mtu * g.qt_magic) >> 20), tq = mtu - b * qtiles;
                epi_tile<EPI, false>(g, j ? acc1[t4] : acc0[t4], mt * 32 + lane31, mtu, b, tq, tq * 32 + lane31, hi, n32, e, ln_a[t4], ln_nb[t4]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3. LayerNorm over channels, fp32 in -> fp16 fragment tiles out.  One wavefront per token.
// ---------------------------------------------------------------------------------------------
// A lane owns the float4 chunks c = lane + 64 i, i < LN_CH (D <= 1024): v[4 i .. 4 i + 3].
constexpr int LN_CH = 4;
__device__ __forceinline__ void ln_load(const float* __restrict__ row, int D, float (&v)[4 * LN_CH], bool (&has)[LN_CH]) {
    const int lane = lane_id(), nchunk = D >> 2;
#pragma unroll
    for (int i = 0; i < LN_CH; ++i) {
        const int c = lane + 64 * i;
        has[i] = c < nchunk;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has[i]) t = *reinterpret_cast<const float4*>(row + c * 4);
        v[4 * i + 0] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
}

__device__ __forceinline__ void wave_ln_stats(const float (&v)[4 * LN_CH], const bool (&has)[LN_CH], int D, float& mean, float& rstd,
                                              float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4 * LN_CH; ++i) s += has[i >> 2] ? v[i] : 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4 * LN_CH; ++i) {
        const float c = has[i >> 2] ? v[i] - mean : 0.f;
        q += c * c;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    rstd = rsqrtf(q / (float)D + eps);
}

__global__ __launch_bounds__(256) void vit_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bsh, int M, int D,
                                                            _Float16* __restrict__ out, int xcd_map) {
    int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (xcd_map) {   // token tile mt on XCD mt % 8: item = 32 lt + token of the tile
        const int it = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
        m = ((blockIdx.x & 7) + 8 * (it >> 5)) * 32 + (it & 31);
    }
    if (m >= M) return;
    const int lane = lane_id();
    float v[4 * LN_CH];
    bool has[LN_CH];
    ln_load(x + (size_t)m * D, D, v, has);
    float mean, rstd;
    wave_ln_stats(v, has, D, mean, rstd, 1e-6f);
#pragma unroll
    for (int i = 0; i < LN_CH; ++i) {
        if (!has[i]) continue;
        const int n0 = (lane + 64 * i) * 4;
        half4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (_Float16)((v[4 * i + j] - mean) * rstd * w[n0 + j] + bsh[n0 + j]);
        *reinterpret_cast<half4*>(out + frag_index(m, n0, D / 16)) = o;
    }
}

// final: LayerNorm(norm, 1e-6) -> drop cls -> LayerNorm(channel norm, 1e-5) -> fp32 [B][Np][D]
__global__ __launch_bounds__(256) void vit_final_kernel(const float* __restrict__ x, const float* __restrict__ nw,
                                                        const float* __restrict__ nb, const float* __restrict__ cw,
                                                        const float* __restrict__ cb, Dims d, float* __restrict__ out) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);  // over B * Np
    if (tok >= d.B * d.Np) return;
    const int b = tok / d.Np, p = tok % d.Np;
    const int m = b * d.Tp + 1 + p;
    const int lane = lane_id();
    float v[4 * LN_CH];
    bool has[LN_CH];
    ln_load(x + (size_t)m * d.D, d.D, v, has);
    float mean, rstd;
    wave_ln_stats(v, has, d.D, mean, rstd, 1e-6f);
#pragma unroll
    for (int i = 0; i < LN_CH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = (lane + 64 * i) * 4 + j;
            if (has[i]) v[4 * i + j] = (v[4 * i + j] - mean) * rstd * nw[n] + nb[n];
        }
    wave_ln_stats(v, has, d.D, mean, rstd, 1e-5f);
#pragma unroll
    for (int i = 0; i < LN_CH; ++i) {
        if (!has[i]) continue;
        const int n0 = (lane + 64 * i) * 4;
        float4 o;
        o.x = (v[4 * i + 0] - mean) * rstd * cw[n0 + 0] + cb[n0 + 0];
        o.y = (v[4 * i + 1] - mean) * rstd * cw[n0 + 1] + cb[n0 + 1];
        o.z = (v[4 * i + 2] - mean) * rstd * cw[n0 + 2] + cb[n0 + 2];
        o.w = (v[4 * i + 3] - mean) * rstd * cw[n0 + 3] + cb[n0 + 3];
        *reinterpret_cast<float4*>(out + (size_t)tok * d.D + n0) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// 4. attention: one wavefront per 32 queries of one (image, head).  NKT = key tiles (Tp / 32).
// ---------------------------------------------------------------------------------------------
// softmax over the keys of NKT score tiles S^T (rows = keys, column = query = lane & 31; the two half-waves hold a query's other keys):
// p = exp2((s - max s) scale) with scale = log2(e) / 8 -- the maximum on the raw scores (v_max3), scale and subtraction as ONE packed FMA per
// pair of scores, the sum by packed adds (round 5: 6 instructions per score were 4 of the kernel's 5 600 VALU cycles per wave; the exponential
// itself -- a quarter-rate instruction -- is the floor).  Keys >= T (padding: only in the last key tile, Tp - T < 32) get -3e38: exp2 gives 0.
// Shared by the two attention kernels (bit-equal to each other).  Returns 1 / sum.
template <int NKT>
__device__ __forceinline__ float att_softmax(floatx16 (&S)[NKT], int T, int hi) {
    const float scale = 0.125f * 1.44269504088896340736f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = (NKT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        S[NKT - 1][r] = (key < T) ? S[NKT - 1][r] : -3.0e38f;
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, S[kt][r]), S[kt][r + 1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const f2 sc2 = splat2(scale), off2 = splat2(-(mx * scale));
    f2 sum2 = splat2(0.f);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f2 a = fma2(f2{S[kt][r], S[kt][r + 1]}, sc2, off2);
            const f2 pr = f2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};   // (v_exp_f32: arguments <= 0, a result below 2^-126 is 0 either way)
            S[kt][r] = pr[0];
            S[kt][r + 1] = pr[1];
            sum2 += pr;
        }
    float sum = sum2[0] + sum2[1];
    sum += __shfl_xor(sum, 32);
    return 1.0f / sum;
}

template <int NKT>
__global__ __launch_bounds__(256) void vit_attention_kernel(const uint4* __restrict__ Q, const uint4* __restrict__ K,
                                                            const uint4* __restrict__ VT, int T, int Tp, int heads, int D,
                                                            int nwork, _Float16* __restrict__ out, int xcd_map) {
    const int lane = lane_id();
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int qtiles = Tp / 32;
    int bh = wid / qtiles, qt = wid % qtiles;
    if (xcd_map) {   // the (token tile mt = b * qtiles + qt, head) pairs of XCD mt % 8
        int mt, head_;
        if (!xcd_item(nwork / heads, heads, threadIdx.x >> 6, mt, head_)) return;
        bh = (mt / qtiles) * heads + head_;
        qt = mt % qtiles;
    } else if (wid >= nwork) return;
    const int b = bh / heads, head = bh % heads;
    const size_t base = (size_t)bh * Tp * 64 / 8;  // uint4 units per (b, head)
    // Q^T as B operand: query tile qt, 4 k-steps over d
    half8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        uint4 v = Q[base + ((size_t)qt * 4 + s) * 64 + lane];
        qf[s] = *reinterpret_cast<half8*>(&v);
    }
    // S^T tiles: rows = keys, column = query (lane & 31)
    floatx16 S[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 v = K[base + ((size_t)kt * 4 + s) * 64 + lane];
            S[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&v), qf[s], S[kt], 0, 0, 0);
        }
    }
    const int hi = lane >> 5;
    // softmax over keys (scale 1/8), keys >= T masked: S becomes the unnormalised probabilities, inv = 1 / their sum
    const float inv = att_softmax<NKT>(S, T, hi);
    // O^T[d][query] = sum_keys V^T[d][key] P^T[key][query]: A = V^T fragment, B = P^T in registers.
    floatx16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }
    const int vks = Tp / 16;  // k-steps over keys
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            // keys 16*(2kt+s2) .. +15: register groups g = 2*s2, 2*s2+1 of tile kt.
            // lanes 0-31 hold keys 8g..8g+3, lanes 32-63 keys 8g+4..8g+7 of each group (packed 2x fp16x2)
            unsigned x0 = pack_f16x2(S[kt][8 * s2 + 0], S[kt][8 * s2 + 1]);
            unsigned x1 = pack_f16x2(S[kt][8 * s2 + 2], S[kt][8 * s2 + 3]);
            unsigned y0 = pack_f16x2(S[kt][8 * s2 + 4], S[kt][8 * s2 + 5]);
            unsigned y1 = pack_f16x2(S[kt][8 * s2 + 6], S[kt][8 * s2 + 7]);
            // swap upper half of x with lower half of y: lower lanes end with keys 16s..16s+7,
            // upper lanes with keys 16s+8..16s+15 -- the B-operand layout (k = 8*(lane>>5) + e)
            auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
            uint4 pb;
            pb.x = r0[0]; pb.y = r1[0]; pb.z = r0[1]; pb.w = r1[1];
            const half8 pf = *reinterpret_cast<half8*>(&pb);
            const int ks = kt * 2 + s2;
            uint4 v0 = VT[base + ((size_t)0 * vks + ks) * 64 + lane];
            uint4 v1 = VT[base + ((size_t)1 * vks + ks) * 64 + lane];
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&v0), pf, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&v1), pf, O1, 0, 0, 0);
        }
    }
    // write O[query][head*64 + d] as fragment tiles of the [M][D] activation for the proj GEMM
    const int t = qt * 32 + (lane & 31);
    const int m = b * Tp + t;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const floatx16& O = half ? O1 : O0;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int dd = half * 32 + 8 * grp + 4 * hi;
            half4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (_Float16)(O[grp * 4 + j] * inv);
            *reinterpret_cast<half4*>(out + frag_index(m, head * 64 + dd, D / 16)) = o;
        }
    }
}

// ---- the same attention with K and V^T of an (image, head) shared by four query tiles through the LDS
// (__launch_bounds__(256, 2): unbounded, the compiler took 432 registers at 11 key tiles -- 176 of them AGPRs the scores were shuttled
// through --; bounded it is 230 VGPRs, no AGPR, no spill, and fewer instructions: 96 images 4.10 -> 3.96 ms)
template <int NKT>
__global__ __launch_bounds__(256, 2) void vit_attention_lds_kernel(const uint4* __restrict__ Q, const uint4* __restrict__ K,
                                                                const uint4* __restrict__ VT, int T, int Tp, int heads, int D,
                                                                _Float16* __restrict__ out) {
    // The four waves of a workgroup take four query tiles of ONE (image, head) and share its K and V^T through the LDS: both are
    // NKT x 4 KiB of consecutive fragment rows in global memory (8 KiB per key tile together: 88 KiB at 352 tokens), copied once per
    // workgroup by LDS-DMA.  The one-wave-per-tile kernel above fetches every fragment from the L2 in every wave: 11 waves per
    // (image, head) read the same 88 KiB -- 545 MB per layer at 96 images, what that kernel's 93 us consist of.
    extern __shared__ __attribute__((aligned(16))) unsigned char att_lds[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qtiles = Tp / 32, ngroups = (qtiles + 3) / 4;
    const int bh = blockIdx.x / ngroups, qt = (blockIdx.x % ngroups) * 4 + wave;
    const int b = bh / heads, head = bh % heads;
    const size_t base = (size_t)bh * Tp * 64 / 8;  // uint4 units per (b, head)
    {
        const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)att_lds;
        // 1 KiB pieces: NKT x 4 of K, then as many of V^T; wave w issues pieces w, w + 4, ... of each -- NKT per wave and operand, K first,
        // so that "at most NKT of my loads in flight" means this wave's share of K has landed while V^T is still on its way
        // (the wave's own query tile goes the same way, first: a compiler-tracked load would be waited for with vmcnt(0) at its first
        // use -- the compiler does not see the DMA behind it -- and V^T would be waited for with it)
        const int qt_ = qt < qtiles ? qt : qtiles - 1;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
            vit_glds16(Q + base + ((size_t)qt_ * 4 + s4) * 64 + lane,
                       __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(NKT * 4 + wave * 4 + s4) * 1024u));
#pragma unroll
        for (int i = 0; i < NKT; ++i)
            vit_glds16(K + base + (size_t)(wave + 4 * i) * 64 + lane, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave + 4 * i) * 1024u));
    }
    // Round 5: V^T takes K's place.  With both resident a workgroup held 104 KiB of LDS -- one workgroup of four waves per compute unit,
    // and the kernel ran at the pace of its staging (173 MB fetched per launch at 96 images, L2 hit rate 0.12, 3.5 TB/s: memory-bound with
    // nothing to overlap the fetch with).  K is dead once the scores are in registers: V^T is copied over it behind a barrier and lands
    // under the softmax, as before; 60 KiB per workgroup, two workgroups per compute unit, one's staging under the other's MFMAs.
    const uint4* K_l = reinterpret_cast<const uint4*>(att_lds);
    const uint4* VT_l = reinterpret_cast<const uint4*>(att_lds);
    const bool active = qt < qtiles;   // (the last group of an image may have fewer than four tiles: those waves only stage)
    // Q^T as B operand: query tile qt, 4 k-steps over d
    vit_wait_vmcnt<0>();   // the query tile and K are here
    __syncthreads();
    half8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        uint4 v = reinterpret_cast<const uint4*>(att_lds + (size_t)(NKT * 4 + wave * 4 + s) * 1024)[lane];
        qf[s] = *reinterpret_cast<half8*>(&v);
    }
    // S^T tiles: rows = keys, column = query (lane & 31)
    floatx16 S[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 v = K_l[((size_t)kt * 4 + s) * 64 + lane];
            S[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&v), qf[s], S[kt], 0, 0, 0);
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();   // every wave has read its last K fragment
    {
        const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)att_lds;
#pragma unroll
        for (int i = 0; i < NKT; ++i)
            vit_glds16(VT + base + (size_t)(wave + 4 * i) * 64 + lane, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave + 4 * i) * 1024u));
    }
    const int hi = lane >> 5;
    // softmax over keys (scale 1/8), keys >= T masked: S becomes the unnormalised probabilities, inv = 1 / their sum
    const float inv = att_softmax<NKT>(S, T, hi);
    vit_wait_vmcnt<0>();   // V^T
    __syncthreads();
    if (!active) return;
    // O^T[d][query] = sum_keys V^T[d][key] P^T[key][query]: A = V^T fragment, B = P^T in registers.
    floatx16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }
    const int vks = Tp / 16;  // k-steps over keys
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            // keys 16*(2kt+s2) .. +15: register groups g = 2*s2, 2*s2+1 of tile kt.
            // lanes 0-31 hold keys 8g..8g+3, lanes 32-63 keys 8g+4..8g+7 of each group (packed 2x fp16x2)
            unsigned x0 = pack_f16x2(S[kt][8 * s2 + 0], S[kt][8 * s2 + 1]);
            unsigned x1 = pack_f16x2(S[kt][8 * s2 + 2], S[kt][8 * s2 + 3]);
            unsigned y0 = pack_f16x2(S[kt][8 * s2 + 4], S[kt][8 * s2 + 5]);
            unsigned y1 = pack_f16x2(S[kt][8 * s2 + 6], S[kt][8 * s2 + 7]);
            // swap upper half of x with lower half of y: lower lanes end with keys 16s..16s+7,
            // upper lanes with keys 16s+8..16s+15 -- the B-operand layout (k = 8*(lane>>5) + e)
            auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
            uint4 pb;
            pb.x = r0[0]; pb.y = r1[0]; pb.z = r0[1]; pb.w = r1[1];
            const half8 pf = *reinterpret_cast<half8*>(&pb);
            const int ks = kt * 2 + s2;
            uint4 v0 = VT_l[((size_t)0 * vks + ks) * 64 + lane];
            uint4 v1 = VT_l[((size_t)1 * vks + ks) * 64 + lane];
            O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&v0), pf, O0, 0, 0, 0);
            O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&v1), pf, O1, 0, 0, 0);
        }
    }
    // write O[query][head*64 + d] as fragment tiles of the [M][D] activation for the proj GEMM
    const int t = qt * 32 + (lane & 31);
    const int m = b * Tp + t;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const floatx16& O = half ? O1 : O0;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int dd = half * 32 + 8 * grp + 4 * hi;
            half4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (_Float16)(O[grp * 4 + j] * inv);
            *reinterpret_cast<half4*>(out + frag_index(m, head * 64 + dd, D / 16)) = o;
        }
    }
}

// ---- QKV + attention of ONE (image, head) in one workgroup (round 6).  The two kernels it replaces hand q, k and V^T of
// every token through HBM: 73 MB written and 73 MB read per layer at 90 images, of the 680 MB a layer moves.  Here they never leave the
// compute unit.  Four waves, one per SIMD, up to 512 registers each:
//   * wave w keeps the token tiles w, w + 4, w + 8 of the image STATIONARY in registers (24 k-steps x 4 registers each: 288) through three
//     products -- with the head's 64 q, 64 k and 64 v rows of the folded weight, 48 KiB each, staged through two LDS slots by LDS-DMA (the
//     next product's weights land under the current one's MFMAs).  One 1 KiB fragment read feeds three MFMAs (the token-stationary GEMM: one);
//   * q goes from the accumulators to the B-operand layout in registers (v_permlane32_swap, as the probabilities do in the attention);
//     K goes to the LDS in the fragment layout the score product reads (the stores of epi_tile, into the LDS);  V is computed as
//     tokens x channels (operands swapped), so that a lane holds 4 consecutive KEYS of one channel: V^T leaves with the same 8-byte stores,
//     no 2-byte scatter;  V^T takes the k weights' slot;
//   * then the attention of vit_attention_lds_kernel for the wave's own query tiles, K and V^T from the LDS.
// The six heads of an image read the same 264 KB of tokens: workgroups of one image are neighbours on ONE XCD (blockIdx % 8), so five of
// the six reads hit that XCD's L2 (the item mapping at the top of the kernel).  D = 384 and at most 12 token tiles (vfm_vit_forward falls back to the two kernels otherwise).
// Same MFMAs over the same fragments in the same k order, the same epilogue arithmetic: q, K, V^T and the output are bit-equal to the two
// kernels' (tests/test_gpu_vit.py).
template <int NKT>
__global__ __launch_bounds__(256, 1) void vit_qkv_attention_kernel(GemmArgs g, int B, _Float16* __restrict__ out) {
    constexpr int KS = 24;                 // k-steps of D = 384
    constexpr int NTW = (NKT + 3) / 4;     // token tiles per wave
    constexpr unsigned SLOT = 24u * 1024u, KR = 2u * SLOT, VR = KR + (unsigned)NKT * 4096u, STR = VR + (unsigned)NKT * 4096u;
    extern __shared__ __attribute__((aligned(16))) unsigned char fl[];   // [2 weight slots x 24 KiB][K: NKT x 4 KiB][V^T: NKT x 4 KiB][(a, nb) per token]
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int lane = lane_id(), hi = lane >> 5, lane31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (image, head) items in image-major order, an eighth of them per XCD (workgroup i runs on XCD i % 8): the heads of an image are
    // neighbours on one XCD -- at most one image per XCD boundary is split --, and every XCD gets the same number of workgroups (whole
    // images per XCD: 84 images = 66 workgroups on the XCDs with 11 images, a third round on 32 compute units for 2 of them)
    const int xcd = blockIdx.x & 7, per_xcd = (int)(gridDim.x >> 3);
    const int item = xcd * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= B * g.heads) return;   // workgroup-uniform
    const int b = item / g.heads, head = item - b * g.heads;
    unsigned long long tr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // (tools: phase times of the workgroup, 100 MHz ticks)
    if (g.dbg) tr[0] = wall_clock64();
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)fl;
    // The six 32-channel weight tiles of the head in the order k0 k1 v0 v1 q0 q1 (q last: its result stays in registers, and by then the
    // tokens' registers are on their way out) through a ring of two 24 KiB slots: 24 consecutive 1 KiB fragment rows each, six per wave.
    auto stage_w = [&](int n) __attribute__((always_inline)) {
        const int which = n < 2 ? 1 : (n < 4 ? 2 : 0), c = n & 1;
        const uint4* src = g.W + (size_t)((which * g.heads + head) * 2 + c) * KS * 64 + lane;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int p = wave + 4 * i;
            vit_glds16(src + (size_t)p * 64, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(n & 1) * SLOT + (unsigned)p * 1024u));
        }
    };
    stage_w(0);
    stage_w(1);
    u32x4 tok[NTW][KS];
    float ln_a[NTW], ln_nb[NTW];
    float2* st_l = reinterpret_cast<float2*>(fl + STR);
    // Issue order = completion order: weights of tiles 0 and 1, the tokens' LayerNorm sums, then the token fragments k-step by k-step -- the
    // first product starts on the first k-steps while the rest of the wave's 72 KiB is on its way (the compiler's own vmcnt before each
    // fragment's first use; waited for as a whole, the loads were 4.9 us of a workgroup's 32 with every compute unit asking at once).
    float4 sraw[NTW][6];
    const u32x4* Ap[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int tile = wave + 4 * i;
        const int tl = tile < NKT ? tile : NKT - 1;   // (a tile past the image: this wave multiplies the last one again and drops the result)
        Ap[i] = reinterpret_cast<const u32x4*>(g.A) + (size_t)(b * NKT + tl) * KS * 64 + lane;
        const float4* sp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(g.stats) + (unsigned)((b * NKT + tl) * 32 + lane31) * 12u * 8u);
#pragma unroll
        for (int k = 0; k < 6; ++k) sraw[i][k] = sp[k];
    }
    // (a compute unit's vector memory path takes 64 B per clock: the 288 KiB of its four waves' fragments are 4 600 cycles of ISSUE, in
    // order, before a wave's first MFMA if they are all requested up front -- the first PRE k-steps here, the rest from inside the first
    // product's k-loop, PRE k-steps ahead)
    constexpr int PRE = 6;
#pragma unroll
    for (int s = 0; s < PRE; ++s)
#pragma unroll
        for (int i = 0; i < NTW; ++i) tok[i][s] = Ap[i][(size_t)s * 64];
    // Where the stationary fragments live is not left to the register allocator (it split them between the two files in a way that
    // needed 76 spills, reloaded from scratch inside the k-loop): tiles 0 and 1 in accumulation registers -- an MFMA reads its operands from
    // either file --, tile 2 and everything the VALU touches in the architectural ones.  (Pinned at their first use, in tile 0's k-loop.)
    if (g.dbg) tr[1] = wall_clock64();
    const unsigned lane_off = (unsigned)lane31 * 16u + 8u * (unsigned)hi;
    floatx16 acc[NTW];
    uint4 qf[NTW][4];   // q as the B operand of the score product: 4 k-steps over the head's 64 channels
    // (the tokens' (a, nb) from the sums requested first: consumed behind the first product's k-loop, whose start they would hold up)
    auto finish_stats = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {   // (ln_stats_load's arithmetic on the sums of D / 32 = 12 slices)
            float sx = 0.f, sq = 0.f;
    #pragma unroll
            for (int k = 0; k < 6; ++k) {
                sx += sraw[i][k].x; sq += sraw[i][k].y;
                sx += sraw[i][k].z; sq += sraw[i][k].w;
            }
            const float mean = sx * g.invD;
            const float var = fmaxf(__builtin_fmaf(sq, g.invD, -(mean * mean)), 0.0f);
            ln_a[i] = __builtin_amdgcn_rsqf(var + 1e-6f);
            ln_nb[i] = -(ln_a[i] * mean);
            const int tile = wave + 4 * i;
            if (tile < NKT && hi == 0) st_l[tile * 32 + lane31] = make_float2(ln_a[i], ln_nb[i]);
        }
};
    // One wave per SIMD: nobody covers an LDS round trip (~130 cycles against the 96 of a k-step's three MFMAs), so the weight fragments
    // come through a ring of PF registers, read PF k-steps ahead (left to the compiler the loop was read - wait - multiply).
    auto run_tile = [&](auto Nc) __attribute__((always_inline)) {
        constexpr int n = decltype(Nc)::value;
        constexpr int which = n < 2 ? 1 : (n < 4 ? 2 : 0), c = n & 1;
        constexpr bool SW = which == 2;   // V as tokens x channels
        // tile n has landed (this wave's pieces: issued a tile ago; the others': the barrier); everybody has left tile n - 1: its slot takes tile n + 1
        if constexpr (n == 0) vit_wait_vmcnt<36>();   // (the weights of tiles 0 and 1: everything older than the 18 loads of sums and the PRE k-steps of fragments)
        else vit_wait_vmcnt<0>();
        __syncthreads();
        if constexpr (n == 3) { if (g.dbg) tr[10] = wall_clock64(); }
        if constexpr (n == 4) { if (g.dbg) tr[13] = wall_clock64(); }
        if constexpr (n >= 1 && n + 1 < 6) stage_w(n + 1);
        // the epilogue's per-channel operands, requested here: an L2 round trip behind the k-loop was 0.5 us per tile with nothing beside it
        const int n32 = __builtin_amdgcn_readfirstlane((which * g.heads + head) * 2 + c);
        float4 bias[4], csum[4];   // (q, K: the tile's 32 channels in the accumulator's row order, epi_load's; V: the lane's channel in [0].x)
        if constexpr (SW) {
            bias[0].x = g.bias[n32 * 32 + lane31];
            csum[0].x = g.csum[n32 * 32 + lane31];
        } else {
            const char* bias_b = reinterpret_cast<const char*>(g.bias + n32 * 32);
            const char* csum_b = reinterpret_cast<const char*>(g.csum + n32 * 32);
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                bias[grp] = *reinterpret_cast<const float4*>(bias_b + (16u * (unsigned)hi + 32u * grp));
                csum[grp] = *reinterpret_cast<const float4*>(csum_b + (16u * (unsigned)hi + 32u * grp));
            }
        }
        constexpr int PF = 4;
        const unsigned char* wp = fl + (unsigned)(n & 1) * SLOT + (unsigned)lane * 16u;
        half8 wr[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) wr[i] = *reinterpret_cast<const half8*>(wp + (unsigned)i * 1024u);
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const half8 wv = wr[s % PF];
            if constexpr (n == 0) {
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    if (i < 2) asm volatile("" : "+a"(tok[i][s]));
                    else asm volatile("" : "+v"(tok[i][s]));
                }
            }
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const half8 tv = *reinterpret_cast<const half8*>(&tok[i][s]);
                if constexpr (SW) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tv, wv, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv, tv, acc[i], 0, 0, 0);
            }
            if (s + PF < KS) wr[s % PF] = *reinterpret_cast<const half8*>(wp + (unsigned)(s + PF) * 1024u);
            if constexpr (n == 0) {
                if (s + PRE < KS) {
#pragma unroll
                    for (int i = 0; i < NTW; ++i) tok[i][s + PRE] = Ap[i][(size_t)(s + PRE) * 64];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (n == 0) finish_stats();
        if constexpr (n == 3) { if (g.dbg) tr[11] = wall_clock64(); }
        if constexpr (n == 4) { if (g.dbg) tr[14] = wall_clock64(); }
        if constexpr (SW) {
            // V^T: rows = the head's 64 channels (2 tiles), k = keys (2 NKT k-steps); a lane holds channel lane31 and keys 8 grp + 4 hi .. + 3 of a tile
            const f2 B2 = splat2(bias[0].x), C2 = splat2(csum[0].x);
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int tile = wave + 4 * i;
                if (tile >= NKT) continue;
                unsigned char* vb = fl + VR + (unsigned)(c * (2 * NKT) * 2 + tile * 4) * 512u + lane_off;
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    const float4 s01 = *reinterpret_cast<const float4*>(st_l + tile * 32 + 8 * grp + 4 * hi);       // (a, nb) of keys + 0, + 1
                    const float4 s23 = *reinterpret_cast<const float4*>(st_l + tile * 32 + 8 * grp + 4 * hi + 2);   // + 2, + 3
                    const f2 v01 = fma2(f2{acc[i][grp * 4 + 0], acc[i][grp * 4 + 1]}, f2{s01.x, s01.z}, fma2(f2{s01.y, s01.w}, C2, B2));
                    const f2 v23 = fma2(f2{acc[i][grp * 4 + 2], acc[i][grp * 4 + 3]}, f2{s23.x, s23.z}, fma2(f2{s23.y, s23.w}, C2, B2));
                    *reinterpret_cast<half4*>(vb + 512u * grp) = to_half4(v01, v23);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int tile = wave + 4 * i;
                const f2 A2 = splat2(ln_a[i]), NB2 = splat2(ln_nb[i]);
                f2 v01[4], v23[4];
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    v01[grp] = fma2(f2{acc[i][grp * 4 + 0], acc[i][grp * 4 + 1]}, A2, fma2(NB2, f2{csum[grp].x, csum[grp].y}, f2{bias[grp].x, bias[grp].y}));
                    v23[grp] = fma2(f2{acc[i][grp * 4 + 2], acc[i][grp * 4 + 3]}, A2, fma2(NB2, f2{csum[grp].z, csum[grp].w}, f2{bias[grp].z, bias[grp].w}));
                }
                if constexpr (which == 1) {
                    // K: fragment rows [key tile][4 k-steps][2 halves][32 keys] x 8 channels, 4 KiB per key tile (epi_tile's stores, into the LDS)
                    if (tile < NKT) {
                        unsigned char* kb = fl + KR + (unsigned)(tile * 8 + c * 4) * 512u + lane_off;
#pragma unroll
                        for (int grp = 0; grp < 4; ++grp) *reinterpret_cast<half4*>(kb + 512u * grp) = to_half4(v01[grp], v23[grp]);
                    }
                } else {
                    unsigned pk[8];
#pragma unroll
                    for (int grp = 0; grp < 4; ++grp) {
                        pk[grp * 2 + 0] = pack_f16x2(v01[grp][0], v01[grp][1]);
                        pk[grp * 2 + 1] = pack_f16x2(v23[grp][0], v23[grp][1]);
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {   // channels 16 s2 .. + 15 of the tile: groups 2 s2 (lower lanes' half) and 2 s2 + 1 (upper lanes')
                        auto r0 = __builtin_amdgcn_permlane32_swap(pk[4 * s2 + 0], pk[4 * s2 + 2], false, false);
                        auto r1 = __builtin_amdgcn_permlane32_swap(pk[4 * s2 + 1], pk[4 * s2 + 3], false, false);
                        qf[i][c * 2 + s2] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                    }
                }
            }
        }
    };
    run_tile(std::integral_constant<int, 0>{});
    run_tile(std::integral_constant<int, 1>{});
    if (g.dbg) tr[2] = wall_clock64();
    run_tile(std::integral_constant<int, 2>{});
    run_tile(std::integral_constant<int, 3>{});
    if (g.dbg) tr[3] = wall_clock64();
    run_tile(std::integral_constant<int, 4>{});
    if (g.dbg) tr[15] = wall_clock64();
    run_tile(std::integral_constant<int, 5>{});
    __syncthreads();   // K and V^T are complete
    if (g.dbg) tr[4] = wall_clock64();
    // ---- attention of the wave's query tiles (vit_attention_lds_kernel's body)
    const uint4* K_l = reinterpret_cast<const uint4*>(fl + KR);
    const uint4* VT_l = reinterpret_cast<const uint4*>(fl + VR);
    constexpr int vks = 2 * NKT;
    // K is the same for the wave's three query tiles and the token fragments' registers are free: every wave keeps ALL of K (4 NKT fragments,
    // 176 registers at 11 key tiles) in registers, read from the LDS once -- with a fragment read per MFMA in all four waves at once the
    // score products ran at the LDS port's pace, twice their MFMAs' time.  V^T still streams (K + V^T + the scores are 528 registers).
    u32x4 kreg[4 * NKT];
#pragma unroll
    for (int p = 0; p < 4 * NKT; ++p) kreg[p] = reinterpret_cast<const u32x4*>(K_l)[(size_t)p * 64 + lane];
#pragma unroll
    for (int p = 0; p < 4 * NKT; ++p) asm volatile("" : "+a"(kreg[p]));
#pragma unroll
    for (int i = 0; i < NTW; ++i) {   // (unrolled: indexed by a loop counter, qf went to the stack)
        const int qt = wave + 4 * i;
        if (qt >= NKT) break;
        uint4 q4[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) q4[s] = qf[i][s];
        floatx16 S[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                S[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&kreg[kt * 4 + s]), *reinterpret_cast<half8*>(&q4[s]), S[kt], 0, 0, 0);
        }
        if (g.dbg && i == 0) tr[7] = wall_clock64();
        const float inv = att_softmax<NKT>(S, g.T, hi);
        if (g.dbg && i == 0) tr[8] = wall_clock64();
        floatx16 O0, O1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }
        {
            constexpr int PFV = 3;
            uint4 vr0[PFV], vr1[PFV];
#pragma unroll
            for (int p = 0; p < PFV; ++p) {
                vr0[p] = VT_l[((size_t)0 * vks + (p < vks ? p : vks - 1)) * 64 + lane];
                vr1[p] = VT_l[((size_t)1 * vks + (p < vks ? p : vks - 1)) * 64 + lane];
            }
            // probabilities of keys 16 ks .. + 15 as the B operand (see vit_attention_kernel), one k-step AHEAD of the MFMAs that take them: the
            // conversions and lane swaps of step ks + 1 issue between the two MFMAs of step ks (one wave per SIMD: nobody else fills that time)
            auto make_pf = [&](int ks) __attribute__((always_inline)) {
                const int kt = ks >> 1, s2 = ks & 1;
                unsigned x0 = pack_f16x2(S[kt][8 * s2 + 0], S[kt][8 * s2 + 1]);
                unsigned x1 = pack_f16x2(S[kt][8 * s2 + 2], S[kt][8 * s2 + 3]);
                unsigned y0 = pack_f16x2(S[kt][8 * s2 + 4], S[kt][8 * s2 + 5]);
                unsigned y1 = pack_f16x2(S[kt][8 * s2 + 6], S[kt][8 * s2 + 7]);
                auto r0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                uint4 pb;
                pb.x = r0[0]; pb.y = r1[0]; pb.z = r0[1]; pb.w = r1[1];
                return pb;
            };
            uint4 pnext = make_pf(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < vks; ++ks) {
                uint4 pcur = pnext;
                const half8 pf = *reinterpret_cast<half8*>(&pcur);
                O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&vr0[ks % PFV]), pf, O0, 0, 0, 0);
                if (ks + 1 < vks) pnext = make_pf(ks + 1);
                O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&vr1[ks % PFV]), pf, O1, 0, 0, 0);
                if (ks + PFV < vks) {
                    vr0[ks % PFV] = VT_l[((size_t)0 * vks + ks + PFV) * 64 + lane];
                    vr1[ks % PFV] = VT_l[((size_t)1 * vks + ks + PFV) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (g.dbg && i == 0) tr[9] = wall_clock64();
        // O[query][head 64 + d] into the fragment tiles of the [M][D] activation (frag_index(m, head 64 + half 32 + 8 grp + 4 hi, KS)): one
        // wave-uniform base + the lane's 16 lane31 + 8 hi bytes + 512 (4 half + grp)
        unsigned char* ob = reinterpret_cast<unsigned char*>(out) + ((size_t)(b * NKT + qt) * KS + (size_t)head * 4) * 1024u + lane_off;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const floatx16& O = half ? O1 : O0;
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                half4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (_Float16)(O[grp * 4 + j] * inv);
                *reinterpret_cast<half4*>(ob + 512u * (unsigned)(half * 4 + grp)) = o;
            }
        }
        if (g.dbg && i == 0) tr[5] = wall_clock64();
    }
    if (g.dbg && wave == 0 && lane == 0) {
        tr[6] = wall_clock64();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
#pragma unroll
        for (int k = 0; k < 16; ++k) g.dbg[(size_t)blockIdx.x * 16 + k] = tr[k];
        (void)xcc;
    }
}

struct VitWs {
    float* x;          // [M][D] fp32 residual
    _Float16* a;       // [M][max(D, KP)] fragment tiles (im2col / attention out)
    _Float16* xh;      // [M][D] fragment tiles: fp16 copy of the residual stream (what the QKV / fc1 GEMMs multiply)
    float* stats;      // [M][D / 32][2] LayerNorm partial sums of the residual stream
    _Float16* h;       // [M][mlp] fragment tiles
    _Float16* q;
    _Float16* k;
    _Float16* vt;
    size_t bytes;
};

inline Dims make_dims(const vfm_vit_config* c, int B, int H, int W) {
    Dims d;
    d.B = B; d.H = H; d.W = W;
    d.gh = c->patch_h; d.gw = c->patch_w;
    d.Hr = c->patch * c->patch_h; d.Wr = c->patch * c->patch_w;
    d.Np = d.gh * d.gw; d.T = d.Np + 1; d.Tp = ceil_div(d.T, 32) * 32;
    d.M = B * d.Tp;
    d.D = c->dim; d.heads = c->heads; d.mlp = c->mlp_dim; d.depth = c->depth;
    d.KP = ceil_div(3 * c->patch * c->patch, 32) * 32;
    return d;
}

inline VitWs carve_vit(void* p, const Dims& d) {
    VfmCarver c(p);
    VitWs w;
    const int kmax = d.D > d.KP ? d.D : d.KP;
    w.x = c.take<float>((size_t)d.M * d.D);
    w.a = c.take<_Float16>((size_t)d.M * kmax);
    w.xh = c.take<_Float16>((size_t)d.M * d.D);
    w.stats = c.take<float>((size_t)d.M * (d.D / 32) * 2);
    w.h = c.take<_Float16>((size_t)d.M * d.mlp);
    w.q = c.take<_Float16>((size_t)d.M * d.D);
    w.k = c.take<_Float16>((size_t)d.M * d.D);
    w.vt = c.take<_Float16>((size_t)d.M * d.D);
    w.bytes = c.used();
    return w;
}


template <int EPI, int NT, int PF>
int launch_gemm_cfg(const GemmArgs& g, hipStream_t st) {
    const int waves = (g.M / 32) * (g.N / (32 * NT));
    // waves per workgroup: one -- the waves share nothing, and single waves are spread over more compute units (one scan, N = 384: 648 waves
    // were 168 workgroups); tools/ab_vit_wpw.py: 6 / 12 / 24 images 0.62 / 0.865 / 1.444 ms with four waves per workgroup, 0.62 / 0.845 / 1.425 with one
    const int wpw = vfm_cfg().vit_wpw > 0 ? vfm_cfg().vit_wpw : 1;
    // xcd_map: every XCD gets ceil(tiles / 8) token tiles' worth of workgroups
    const int grid = g.xcd_map ? 8 * ceil_div(ceil_div(g.M / 32, 8) * (g.N / (32 * NT)), wpw) : ceil_div(waves, wpw);
    hipLaunchKernelGGL((vit_gemm_kernel<EPI, NT, PF>), dim3(grid), dim3(64 * wpw), 0, st, g);
    VFM_CHECK_LAUNCH("vit_gemm_kernel");
    return VFM_OK;
}

// Wave tile (NT x 32 channels) and prefetch depth (PF k-steps).  Round 2 measurements on 6 x 1200 x 1600 (tools/ab_vit.py,
// profiles/r02_pmc_vit.json): a wave lives ~2.5 us inside kernels that take 10-17 us -- the forward is bound by the
// ~5 us floor of each of its 87 launches, not by L2 latency or bandwidth: PF = 8 / 16 / 24 make no difference, 32-channel
// tiles everywhere (twice the waves) give 0.76 instead of 0.82 ms, 64 x 64 wave tiles and row-complete workgroups with the
// LayerNorm fused into the epilogue (63 launches, but 66 workgroups per GEMM) gave 1.04 ms and were removed again.
template <int EPI>
int launch_gemm(const GemmArgs& g, hipStream_t st) {
    if constexpr (EPI == EPI_QKV || EPI == EPI_GELU) {
        const int groups = ceil_div(g.M / 32, 4);
        const int nw = vfm_cfg().vit_astat_nw == 112 ? 112 : (vfm_cfg().vit_astat_nw > 0 && (g.N / 32) % vfm_cfg().vit_astat_nw == 0 ? vfm_cfg().vit_astat_nw : 12);
        // One workgroup per compute unit: the kernel pays when its rounds are full.  vfm_cfg().vit_astat_min > 0: from that many groups on whatever
        // the fill (tools); 0 (default): when the last round of `ncu` workgroups is at least three quarters full and there is at least one
        // such round -- 69 ... 93 images of 1200 x 1600 at a time on 256 compute units (tools/ab_vit_astat.py: 48 images 2.19 -> 2.30 ms,
        // 84: 3.35 -> 3.22, 93: 3.60 -> 3.49, 96 = 264 groups, eight of them alone in a second round: 3.95 -> 4.58)
        static std::atomic<int> ncu{0};   // (atomics: vfm_vit_forward may run on two host threads at once -- ViTS14.SPLIT_FROM)
        if (ncu == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            (void)hipGetDevice(&dev);
            ncu = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        const int last_round = groups % ncu;
        const bool full_rounds = groups >= (3 * ncu) / 4 && (last_round == 0 || 4 * last_round >= 3 * ncu);
        const bool want = vfm_cfg().vit_astat_min > 0 ? groups >= vfm_cfg().vit_astat_min : (vfm_cfg().vit_astat_min == 0 && full_rounds);
        if (want && 4 * g.KS * 1024 <= 128 * 1024 && (nw == 112 || (g.N / 32) % nw == 0)) {
#define VIT_ASTAT(NW)                                                                                                                \
    do {                                                                                                                             \
        static std::atomic<unsigned long long> attr_set{0ull};                                                                                   \
        int dev = 0;                                                                                                                 \
        (void)hipGetDevice(&dev);                                                                                                    \
        if (!((attr_set >> (dev & 63)) & 1ull)) {                                                                                    \
            VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_gemm_astat_kernel<EPI, NW, 6>),                     \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));                              \
            attr_set |= 1ull << (dev & 63);                                                                                          \
        }                                                                                                                            \
        GemmArgs ga = g;                                                                                                             \
        ga.dbg = vfm_cfg().vit_trace_fused ? nullptr : vfm_cfg().vit_astat_dbg;                                                                                                    \
        hipLaunchKernelGGL((vit_gemm_astat_kernel<EPI, NW, 6>), dim3(groups), dim3(64 * NW), 4 * g.KS * 1024, st, ga);               \
    } while (0)
            if (vfm_cfg().vit_astat_two && (g.N / 64) >= 8 && g.N % 64 == 0 && g.KS % 4 == 0) {   // two channel tiles per wave, eight waves (round 5)
                static std::atomic<unsigned long long> attr2{0ull};
                int dev2 = 0;
                (void)hipGetDevice(&dev2);
                if (!((attr2 >> (dev2 & 63)) & 1ull)) {
                    VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_gemm_astat2_kernel<EPI, 8, 4>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                    attr2 |= 1ull << (dev2 & 63);
                }
                hipLaunchKernelGGL((vit_gemm_astat2_kernel<EPI, 8, 4>), dim3(groups), dim3(512), 4 * g.KS * 1024, st, g);
                VFM_CHECK_LAUNCH("vit_gemm_astat2_kernel");
                return VFM_OK;
            }
            switch (nw) {
                case 6: VIT_ASTAT(6); break;
                case 112: {   // (A/B) twelve waves, non-temporal output stores
                    static std::atomic<unsigned long long> attr_set2{0ull};
                    int dev = 0;
                    (void)hipGetDevice(&dev);
                    if (!((attr_set2 >> (dev & 63)) & 1ull)) {
                        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_gemm_astat_kernel<EPI, 12, 6, true>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                        attr_set2 |= 1ull << (dev & 63);
                    }
                    GemmArgs ga = g;
                    ga.dbg = vfm_cfg().vit_trace_fused ? nullptr : vfm_cfg().vit_astat_dbg;
                    hipLaunchKernelGGL((vit_gemm_astat_kernel<EPI, 12, 6, true>), dim3(groups), dim3(768), 4 * g.KS * 1024, st, ga);
                } break;
                case 8: VIT_ASTAT(8); break;
                case 16: VIT_ASTAT(16); break;
                default: VIT_ASTAT(12); break;
            }
#undef VIT_ASTAT
            VFM_CHECK_LAUNCH("vit_gemm_astat_kernel");
            return VFM_OK;
        }
    }
    if (g.N % 128 == 0 && vfm_cfg().vit_lds_min_wg > 0 && g.KS % (vfm_cfg().vit_lds_shape / 10 == 4 ? 4 : 2) == 0) {   // (whole stages of 2 or 4 k-steps)
        const int wgs = ceil_div(g.M / 32, 4) * (g.N / 128);
        if (wgs >= vfm_cfg().vit_lds_min_wg) {
            GemmArgs gl = g;
            gl.hot_a = vfm_cfg().vit_hot_a;
            gl.dbg = EPI == EPI_RESID && (vfm_cfg().vit_trace_fused == 0 || (vfm_cfg().vit_trace_fused == 2 && g.KS * 16 == g.D)) ? vfm_cfg().vit_astat_dbg : nullptr;   // (2: the proj launches only)   // (the trace buffer serves whichever kernel a tool looks at)
            if (EPI == EPI_RESID && g.N == 384 && vfm_cfg().vit_wide_tile && g.KS % 2 == 0) {   // one workgroup per 128 tokens x all 384 channels (NG = 3)
                const int gridw = 8 * ceil_div(ceil_div(g.M / 32, 4), 8);
                constexpr int ldsw = 4 * 16 * 2 * 1024;   // NS = 4 stages of (4 + 12) x KB = 2 KiB: 128 KiB
                static std::atomic<unsigned long long> attr_w{0ull};
                int devw = 0;
                (void)hipGetDevice(&devw);
                if (!((attr_w >> (devw & 63)) & 1ull)) {
                    VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_gemm_lds_kernel<EPI, 2, 4, 3, 8>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, ldsw));
                    attr_w |= 1ull << (devw & 63);
                }
                if (gl.dbg) {
                    VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_gemm_lds_kernel<EPI, 2, 4, 3, 8, true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, ldsw));
                    hipLaunchKernelGGL((vit_gemm_lds_kernel<EPI, 2, 4, 3, 8, true>), dim3(gridw), dim3(512), ldsw, st, gl);
                } else
                hipLaunchKernelGGL((vit_gemm_lds_kernel<EPI, 2, 4, 3, 8>), dim3(gridw), dim3(512), ldsw, st, gl);
                VFM_CHECK_LAUNCH("vit_gemm_lds_kernel (128 x 384)");
                return VFM_OK;
            }
            const int grid = 8 * ceil_div(ceil_div(g.M / 32, 4), 8) * (g.N / 128);   // every XCD: ceil(groups / 8) token groups x channel groups
#define VIT_LDS(KB, NS)                                                                                                              \
    do {                                                                                                                             \
        static std::atomic<unsigned long long> attr_set{0ull};                                                                                   \
        int dev = 0;                                                                                                                 \
        (void)hipGetDevice(&dev);                                                                                                    \
        if (!((attr_set >> (dev & 63)) & 1ull)) {                                                                                    \
            VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_gemm_lds_kernel<EPI, KB, NS>),                      \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, NS * 8 * KB * 1024));                      \
            attr_set |= 1ull << (dev & 63);                                                                                          \
        }                                                                                                                            \
        if (gl.dbg) {                                                                                                                \
            VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_gemm_lds_kernel<EPI, KB, NS, 1, 4, true>),          \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, NS * 8 * KB * 1024));                      \
            hipLaunchKernelGGL((vit_gemm_lds_kernel<EPI, KB, NS, 1, 4, true>), dim3(grid), dim3(256), NS * 8 * KB * 1024, st, gl);   \
        } else                                                                                                                       \
        hipLaunchKernelGGL((vit_gemm_lds_kernel<EPI, KB, NS>), dim3(grid), dim3(256), NS * 8 * KB * 1024, st, gl);                    \
    } while (0)
            switch (vfm_cfg().vit_lds_shape) {
                case 42: VIT_LDS(4, 2); break;
                case 43: VIT_LDS(4, 3); break;
                case 24: VIT_LDS(2, 4); break;
                case 25: VIT_LDS(2, 5); break;
                case 26: VIT_LDS(2, 6); break;
                default: VIT_LDS(2, 3); break;
            }
#undef VIT_LDS
            VFM_CHECK_LAUNCH("vit_gemm_lds_kernel");
            return VFM_OK;
        }
    }
    const int cfg = g.N <= 512 ? vfm_cfg().vit_cfg_narrow : vfm_cfg().vit_cfg_wide;
    switch (cfg) {
        case 116: return launch_gemm_cfg<EPI, 1, 16>(g, st);
        case 208: return launch_gemm_cfg<EPI, 2, 8>(g, st);
        default: return launch_gemm_cfg<EPI, 1, 8>(g, st);
    }
}

}  // namespace


VFM_EXPORT size_t vfm_vit_weights_bytes(const vfm_vit_config* cfg) { return cfg ? make_layout(cfg).total : 0; }

// segment table for the host-side packer (vfmreg/vit.py): offsets / sizes in bytes, returns count
VFM_EXPORT int vfm_vit_weights_layout(const vfm_vit_config* cfg, int64_t* offsets_host, int64_t* bytes_host, int max_n) {
    if (!cfg) return 0;
    const Layout L = make_layout(cfg);
    for (int i = 0; i < L.count && i < max_n; ++i) {
        offsets_host[i] = (int64_t)L.off[i];
        bytes_host[i] = (int64_t)L.bytes[i];
    }
    return L.count;
}

VFM_EXPORT size_t vfm_vit_workspace_bytes(const vfm_vit_config* cfg, int B) {
    if (!cfg) return 0;
    return carve_vit(nullptr, make_dims(cfg, B, 1, 1)).bytes;
}

VFM_EXPORT int vfm_vit_forward(const vfm_vit_config* cfg, const void* weights, const uint8_t* img, int B, int H, int W,
                               float* tokens_out, void* ws, size_t ws_bytes, vfm_stream_t stream) {
    VFM_CHECK_ARG(cfg && weights && img && tokens_out && ws, "vit: null pointer");
    VFM_CHECK_ARG(cfg->dim % 64 == 0 && cfg->dim == cfg->heads * 64 && cfg->dim <= 1024, "vit: dim must be heads*64 and <= 1024");
    VFM_CHECK_ARG(cfg->mlp_dim % 64 == 0 && cfg->depth >= 1 && cfg->depth <= 64, "vit: bad mlp_dim / depth");
    VFM_CHECK_ARG(cfg->patch == 14 && cfg->patch_h >= 1 && cfg->patch_w >= 1 && B >= 1, "vit: bad patch grid");
    const Dims d = make_dims(cfg, B, H, W);
    VFM_CHECK_ARG(d.Tp / 32 <= 16, "vit: at most 512 tokens per image supported (got %d)", d.T);
    // (the GEMM epilogues address every activation buffer with 32-bit byte offsets and find a token tile's image by a 20-bit reciprocal)
    VFM_CHECK_ARG((uint64_t)d.M * (uint64_t)(d.mlp > 3 * d.D ? d.mlp : 3 * d.D) * 4u < (1ull << 32) && d.M / 32 < 65536 &&
                  (uint64_t)(d.M / 32) * ((1u << 20) / (unsigned)(d.Tp / 32) + 1u) < (1ull << 32),   // (the reciprocal's product is 32-bit in the kernels)
                  "vit: batch of %d images too large for one call", B);
    if (ws_bytes < carve_vit(nullptr, d).bytes) return vfm_fail(VFM_EWORKSPACE, "vit: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const Layout L = make_layout(cfg);
    const unsigned char* wb = static_cast<const unsigned char*>(weights);
    auto f32 = [&](int seg) { return reinterpret_cast<const float*>(wb + L.off[seg]); };
    auto f16 = [&](int seg) { return reinterpret_cast<const uint4*>(wb + L.off[seg]); };
    VitWs w = carve_vit(ws, d);

    // padded token rows stay exactly zero in the residual stream
    {
        const int64_t total = (int64_t)d.M * (d.KP / 16) * 2;
        if (vfm_cfg().vit_preprocess_patch && d.KP <= 640 && d.W >= 2)
            hipLaunchKernelGGL(vit_preprocess_patch_kernel, dim3((unsigned)d.M), dim3(256), 0, st, img, d, w.a);
        else
            hipLaunchKernelGGL(vit_preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img, d, w.a);
        VFM_CHECK_LAUNCH("vit_preprocess_kernel");
    }
    GemmArgs g{};
    g.T = d.T; g.Tp = d.Tp; g.D = d.D; g.heads = d.heads; g.M = d.M;
    g.x = w.x; g.q = w.q; g.k = w.k; g.vt = w.vt;
    g.xh = w.xh; g.stats = w.stats; g.invD = 1.0f / (float)d.D; g.qt_magic = (1u << 20) / (unsigned)(d.Tp / 32) + 1u;
    g.xcd_map = vfm_cfg().vit_xcd;
    const int att_grid = vfm_cfg().vit_xcd ? 8 * ceil_div(ceil_div(d.M / 32, 8) * d.heads, 4) : ceil_div(d.B * d.heads * (d.Tp / 32), 4);
    // patch embedding (+ cls token + position embedding)
    g.A = reinterpret_cast<const uint4*>(w.a); g.W = f16(SEG_PATCH_W); g.bias = f32(SEG_PATCH_B);
    g.N = d.D; g.KS = d.KP / 16; g.clspos = f32(SEG_CLS_POS);
    int rc = launch_gemm<EPI_PATCH>(g, st);
    if (rc) return rc;
    const int att_work = d.B * d.heads * (d.Tp / 32);
    // K / V^T of an (image, head) shared by four query tiles through the LDS from vfm_cfg().vit_att_lds_min images on (0: never): at one scan the
    // one-wave-per-tile kernel's 396 waves spread over the chip win, at batches the shared form reads a quarter of the L2 bytes
    const bool att_lds = vfm_cfg().vit_att_lds_min > 0 && d.B >= vfm_cfg().vit_att_lds_min && d.Tp / 32 <= 16;
    // QKV + attention per (image, head) in one workgroup (vit_qkv_attention_kernel).  vfm_cfg().vit_fused_qkv: n > 0 from n images on, -1 never,
    // 0 (default) the policy: one workgroup per compute unit, B x heads of them, an eighth per XCD in rounds of 32 -- from 24 images on
    // (below, the two kernels' many small workgroups fill the chip better) unless a second round would be less than a quarter full
    // (profiles/r06_ab_vit_fused_qkv_sweep.txt: 42 images -12.6 %, 44: +1.7 %, 48: +0.2 %, 54: -2.1 %, 84: -11 %, 86: -3.7 %; three
    // rounds and more: -4 ... -9 % wherever the last one ends)
    bool fused_qkv = false;
    if (d.D == 384 && d.Tp / 32 <= 12) {
        const int per_xcd = ceil_div(d.B * d.heads, 8), rounds = ceil_div(per_xcd, 32), last = per_xcd - 32 * (rounds - 1);
        const int k = vfm_cfg().vit_fused_qkv;
        fused_qkv = k > 0 ? d.B >= k : (k == 0 && d.B >= 24 && (rounds >= 3 || rounds == 1 || 4 * last >= 32));
    }
    // fc1 -> GELU -> fc2 in one workgroup per 128 tokens (vit_mlp_kernel).  vfm_cfg().vit_fused_mlp: n > 0 from n images on, -1 never, 0 (default)
    // the policy: one workgroup per compute unit whose loop moves nothing through HBM and whose epilogue moves everything -- a single round runs
    // in lockstep and ends level with the two kernels; from two rounds on the workgroups drift apart and it wins, IF the last round is full
    // (profiles/r06_ab_vit_fused_mlp_sweep.txt: 84 images 0 %, 144: +1.5 %, 156: -1.9 %, 168: -5.3 %, 180: -6.8 %, 192: +5.4 %, 252: -8.9 %)
    bool fused_mlp = false;
    if (d.D == 384 && d.mlp == 1536) {
        static std::atomic<int> ncu_m{0};
        if (ncu_m == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            (void)hipGetDevice(&dev);
            ncu_m = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        const int groups = ceil_div(d.M / 32, 4), rounds = ceil_div(groups, ncu_m), k = vfm_cfg().vit_fused_mlp;
        fused_mlp = k > 0 ? d.B >= k : (k == 0 && rounds >= 2 && 5 * groups >= 4 * rounds * ncu_m);
    }
    for (int l = 0; l < d.depth; ++l) {
        const int s0 = SEG_LAYER0 + l * SEGS_PER_LAYER;
        // LayerNorm 1 is inside this GEMM: raw residual stream x folded weight, statistics applied in the epilogue
        g.A = reinterpret_cast<const uint4*>(w.xh); g.W = f16(s0 + L_QKV_W); g.bias = f32(s0 + L_QKV_B); g.csum = f32(s0 + L_QKV_C);
        g.N = 3 * d.D; g.KS = d.D / 16;
        if (fused_qkv) {
#define VIT_FQA(NKT)                                                                                                      \
    do {                                                                                                                  \
        constexpr int lds_ = 48 * 1024 + 2 * NKT * 4096 + NKT * 32 * 8;                                                       \
        static std::atomic<unsigned long long> attr_set{0ull};                                                            \
        int dev_ = 0;                                                                                                     \
        (void)hipGetDevice(&dev_);                                                                                        \
        if (!((attr_set >> (dev_ & 63)) & 1ull)) {                                                                        \
            VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_qkv_attention_kernel<NKT>),             \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_));                         \
            attr_set |= 1ull << (dev_ & 63);                                                                              \
        }                                                                                                                 \
        GemmArgs gf = g;                                                                                                  \
        gf.dbg = vfm_cfg().vit_trace_fused == 1 ? vfm_cfg().vit_astat_dbg : nullptr;                                                                             \
        hipLaunchKernelGGL(vit_qkv_attention_kernel<NKT>, dim3(8 * ceil_div(d.B * d.heads, 8)), dim3(256), lds_, st, gf, d.B, w.a);       \
    } while (0)
            switch (d.Tp / 32) {
                case 1: VIT_FQA(1); break;   case 2: VIT_FQA(2); break;   case 3: VIT_FQA(3); break;
                case 4: VIT_FQA(4); break;   case 5: VIT_FQA(5); break;   case 6: VIT_FQA(6); break;
                case 7: VIT_FQA(7); break;   case 8: VIT_FQA(8); break;   case 9: VIT_FQA(9); break;
                case 10: VIT_FQA(10); break; case 11: VIT_FQA(11); break;
                default: VIT_FQA(12); break;
            }
#undef VIT_FQA
            VFM_CHECK_LAUNCH("vit_qkv_attention_kernel");
        } else {
        if ((rc = launch_gemm<EPI_QKV>(g, st))) return rc;
#define VIT_ATT(NKT)                                                                                                      \
    if (att_lds) {                                                                                                        \
        static std::atomic<unsigned long long> attr_set{0ull};                                                                        \
        int dev_ = 0;                                                                                                     \
        (void)hipGetDevice(&dev_);                                                                                        \
        if (!((attr_set >> (dev_ & 63)) & 1ull)) {                                                                        \
            VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_attention_lds_kernel<NKT>),             \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, NKT * 4096 + 16384));           \
            attr_set |= 1ull << (dev_ & 63);                                                                              \
        }                                                                                                                 \
        hipLaunchKernelGGL(vit_attention_lds_kernel<NKT>, dim3(d.B * d.heads * ((d.Tp / 32 + 3) / 4)), dim3(256), NKT * 4096 + 16384, st, \
                           reinterpret_cast<const uint4*>(w.q), reinterpret_cast<const uint4*>(w.k),                      \
                           reinterpret_cast<const uint4*>(w.vt), d.T, d.Tp, d.heads, d.D, w.a);                           \
    } else                                                                                                                \
    hipLaunchKernelGGL(vit_attention_kernel<NKT>, dim3(att_grid), dim3(256), 0, st,                                       \
                       reinterpret_cast<const uint4*>(w.q), reinterpret_cast<const uint4*>(w.k),                          \
                       reinterpret_cast<const uint4*>(w.vt), d.T, d.Tp, d.heads, d.D, att_work, w.a, vfm_cfg().vit_xcd)
        switch (d.Tp / 32) {
            case 1: VIT_ATT(1); break;   case 2: VIT_ATT(2); break;   case 3: VIT_ATT(3); break;
            case 4: VIT_ATT(4); break;   case 5: VIT_ATT(5); break;   case 6: VIT_ATT(6); break;
            case 7: VIT_ATT(7); break;   case 8: VIT_ATT(8); break;   case 9: VIT_ATT(9); break;
            case 10: VIT_ATT(10); break; case 11: VIT_ATT(11); break; case 12: VIT_ATT(12); break;
            case 13: VIT_ATT(13); break; case 14: VIT_ATT(14); break; case 15: VIT_ATT(15); break;
            default: VIT_ATT(16); break;
        }
#undef VIT_ATT
        VFM_CHECK_LAUNCH("vit_attention_kernel");
        }
        g.A = reinterpret_cast<const uint4*>(w.a); g.W = f16(s0 + L_PROJ_W); g.bias = f32(s0 + L_PROJ_B);
        g.gamma = f32(s0 + L_LS1); g.N = d.D; g.KS = d.D / 16;
        if ((rc = launch_gemm<EPI_RESID>(g, st))) return rc;
        // LayerNorm 2 likewise
        g.A = reinterpret_cast<const uint4*>(w.xh); g.W = f16(s0 + L_FC1_W); g.bias = f32(s0 + L_FC1_B); g.csum = f32(s0 + L_FC1_C);
        g.N = d.mlp; g.KS = d.D / 16; g.out = w.h;
        if (fused_mlp) {   // fc1 -> GELU -> fc2 of 128 tokens in one workgroup (vit_mlp_kernel, the file's second compile part)
            GemmArgs g2 = g;
            g.dbg = vfm_cfg().vit_trace_fused == 3 ? vfm_cfg().vit_astat_dbg : nullptr;   // (tools: 3 = the trace buffer is vit_mlp_kernel's)
            g2.A = reinterpret_cast<const uint4*>(w.h); g2.W = f16(s0 + L_FC2_W); g2.bias = f32(s0 + L_FC2_B);
            g2.gamma = f32(s0 + L_LS2); g2.N = d.D; g2.KS = d.mlp / 16;
            if ((rc = vfm_vit_launch_mlp_(&g, &g2, st))) return rc;
            continue;
        }
        if ((rc = launch_gemm<EPI_GELU>(g, st))) return rc;
        g.A = reinterpret_cast<const uint4*>(w.h); g.W = f16(s0 + L_FC2_W); g.bias = f32(s0 + L_FC2_B);
        g.gamma = f32(s0 + L_LS2); g.N = d.D; g.KS = d.mlp / 16;
        if ((rc = launch_gemm<EPI_RESID>(g, st))) return rc;
    }
    const int tail = SEG_LAYER0 + d.depth * SEGS_PER_LAYER;
    hipLaunchKernelGGL(vit_final_kernel, dim3(ceil_div(d.B * d.Np, 4)), dim3(256), 0, st, w.x, f32(tail + 0), f32(tail + 1),
                       f32(tail + 2), f32(tail + 3), d, tokens_out);
    VFM_CHECK_LAUNCH("vit_final_kernel");
    return VFM_OK;
}

#else   // VFM_VIT_PART == 1: the fused MLP kernel, compiled without -amdgpu-mfma-vgpr-form

// ---- fc1 -> GELU -> fc2 (+ LayerScale, residual, fp16 copy, LayerNorm sums) of 128 tokens in ONE workgroup (round 6).  The two kernels it
// replaces hand the hidden activations (tokens x 1536 fp16) through HBM: 87 MB written and 87 MB read per layer at 84 images, and fc2 runs
// at the pace of its tiles' staging (545 MB through the L2 -> LDS path per launch).  Here a wave -- one per SIMD, 512 registers -- keeps ONE
// token tile's 24 fragments of the raw residual copy (96 registers) and its 12 output tiles' accumulators (192) for the whole kernel and
// walks the 48 chunks of 32 hidden channels: fc1's 24 MFMAs of the chunk (W1 tile from the LDS), the LayerNorm fold + exact GELU on the 16
// values per lane, the result converted to the B-operand layout in registers (v_permlane32_swap: the q / P conversion), fc2's 24 MFMAs
// (the chunk's two k-steps of all 12 output tiles of W2 from the LDS).  Weights stream through a ring of two 48 KiB LDS slots by LDS-DMA
// (2.36 MB per workgroup, from the L2); the hidden activations never exist outside registers.  Every MFMA takes one 1 KiB fragment from
// the LDS -- the LDS port and the matrix pipes peak together (tools/probe/mfma_lds_probe.hip: ~1.1 PFLOP/s at that ratio) -- against the
// 0.56 PFLOP/s of fc1 + fc2 as two kernels.
// Same MFMAs over the same fragments in the same k order (fc2's accumulators take the hidden channels in ascending order), the same epilogue
// arithmetic (epi_tile): bit-equal to the two kernels (tests/test_gpu_vit.py).  D = 384, mlp = 1536.
__global__ __launch_bounds__(256, 1) void vit_mlp_kernel(GemmArgs g1, GemmArgs g2) {
    constexpr int KS = 24, NC = 48, NO = 12, KS2 = 96;
    constexpr unsigned SLOT = 48u * 1024u, W2OFF = 24u * 1024u, TAB = 3u * SLOT;
    extern __shared__ __attribute__((aligned(16))) unsigned char ml[];   // [3 slots: W1 tile 24 KiB | W2 chunk 24 KiB][bias1 1536 f32][csum1 1536 f32]
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int lane = lane_id(), hi = lane >> 5, lane31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mtiles = g1.M / 32, mg = blockIdx.x;
    int mt = mg * 4 + wave;
    const bool live = mt < mtiles;   // wave-uniform (a partial last group: its waves multiply the last tile again and store nothing)
    if (!live) mt = mtiles - 1;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ml;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, c_wait = 0;   // (tools: start, loop start, loop end in 100 MHz ticks; shader clocks spent at the loop's waits + barriers)
    if (g1.dbg) tr0 = wall_clock64();
    // The loop works on PAIRS of hidden chunks: fc1 of one chunk is ONE accumulator chain of 24 dependent MFMAs, and with two chunks' chains side
    // by side the order A_a A_b B B puts four MFMAs between the links of either.  (Measured: the MFMA-only skeleton went from 2 335 to 2 222
    // shader clocks per chunk -- the chain was not what set the pace; a lone wave issues MFMAs at ~46 clocks each in this loop whatever their
    // order, tools/build_ablate_mlp.sh -- but the halves below are also what lets three 48 KiB slots cover two half-iterations of lead.)
    // Iteration i: fc1 of pair i + 1 (chunks 2 i + 2, 2 i + 3 -> hnA, hnB), LayerNorm fold + GELU of pair i
    // (hcA, hcB -> fout), fc2 of pair i - 1 (fin), interleaved; 96 MFMAs.  Weights by HALVES of an iteration, one 48 KiB slot each, three slots
    // in flight: half H = 2 i + e holds [k-steps 12 e .. + 11 of W1's tiles 2 i + 2 and 2 i + 3: piece 2 kk + ab][W2's chunk 2 i - 2 + e: piece
    // 12 s2 + j], is requested two halves ahead (an L2 round trip of the LDS-DMA is ~2 us when 231 workgroups ask together) and waited for
    // with a counted vmcnt: every half requests exactly twelve pieces per wave (indices past either end are clamped: finite weights nobody's
    // result depends on -- the first iteration's fc2 multiplies zeros).
    auto stage_half = [&](int H, int slot) __attribute__((always_inline)) {
        const int i = H >> 1, e = H & 1;
        const unsigned dst = lds_base + (unsigned)slot * SLOT;
        int ta = 2 * i + 2;
        ta = ta > NC - 2 ? NC - 2 : ta;
        int ch = 2 * i - 2 + e;
        ch = ch < 0 ? 0 : (ch > NC - 1 ? NC - 1 : ch);
        const uint4* s1 = g1.W + ((size_t)ta * KS + (size_t)(12 * e)) * 64 + lane;
        const uint4* s2p = g2.W + (size_t)(2 * ch) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int p = wave + 4 * t, kk = p >> 1, ab = p & 1;
            vit_glds16(s1 + ((size_t)ab * KS + (size_t)kk) * 64, __builtin_amdgcn_readfirstlane(dst + (unsigned)p * 1024u));
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int q = wave + 4 * t, s2 = q >= 12 ? 1 : 0, j = q - 12 * s2;
            vit_glds16(s2p + ((size_t)j * KS2 + (size_t)s2) * 64, __builtin_amdgcn_readfirstlane(dst + W2OFF + (unsigned)q * 1024u));
        }
    };
    // prologue: W1's tiles 0 and 1 as they lie (48 consecutive fragment rows) into slot 2, halves 0 and 1 into slots 0 and 1
    {
        const uint4* s1 = g1.W + lane;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            const int p = wave + 4 * t;
            vit_glds16(s1 + (size_t)p * 64, __builtin_amdgcn_readfirstlane(lds_base + 2u * SLOT + (unsigned)p * 1024u));
        }
    }
    stage_half(0, 0);
    stage_half(1, 1);
    float* tab = reinterpret_cast<float*>(ml + TAB);
    for (int i = threadIdx.x; i < 1536; i += 256) {
        tab[i] = g1.bias[i];
        tab[1536 + i] = g1.csum[i];
    }
    u32x4 xf[KS];
    {
        const u32x4* Ap = reinterpret_cast<const u32x4*>(g1.A) + (size_t)mt * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = Ap[(size_t)s * 64];
    }
    const int m = mt * 32 + lane31;
    float ln_a, ln_nb;
    ln_stats_load(g1, m, ln_a, ln_nb);
    const f2 A2 = splat2(ln_a), NB2 = splat2(ln_nb);
    floatx16 oacc[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.f;
    vit_wait_vmcnt<0>();
    __syncthreads();
    constexpr int PF = 6;
    floatx16 hA0, hB0, hA1, hB1;
    {   // fc1 of pair 0 on its own (slot 2: tile t's k-step k at piece 24 t + k)
        const unsigned char* sl = ml + 2u * SLOT + (unsigned)lane * 16u;
#pragma unroll
        for (int r = 0; r < 16; ++r) { hA0[r] = 0.f; hB0[r] = 0.f; }
        half8 wr[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) wr[i] = *reinterpret_cast<const half8*>(sl + (unsigned)((i & 1) * 24 + (i >> 1)) * 1024u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 2 * KS; ++p) {   // p = 2 k + ab
            const int k = p >> 1;
            if (p & 1) hB0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[p % PF], *reinterpret_cast<const half8*>(&xf[k]), hB0, 0, 0, 0);
            else hA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[p % PF], *reinterpret_cast<const half8*>(&xf[k]), hA0, 0, 0, 0);
            if (p + PF < 2 * KS) wr[p % PF] = *reinterpret_cast<const half8*>(sl + (unsigned)(((p + PF) & 1) * 24 + ((p + PF) >> 1)) * 1024u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long cy1 = 0, cy2 = 0;
    if (g1.dbg) { tr1 = wall_clock64(); cy1 = __builtin_readcyclecounter(); }
    // one half of an iteration: e = 0 / 1 (compile time), i the pair, (sr, sw) = the slot read / the slot half 2 i + e + 2 goes to
    auto half = [&](auto Ec, int i, int sr, int sw, const floatx16& hc, floatx16& hnA, floatx16& hnB, const uint4 (&fin)[4], uint4 (&fout)[4]) __attribute__((always_inline)) {
        constexpr int e = decltype(Ec)::value;
        unsigned long long tw = 0;
        if (g1.dbg) tw = __builtin_readcyclecounter();
        vit_wait_vmcnt<12>();   // everything this wave requested but the last half's twelve pieces has landed
        __syncthreads();        // everybody's; everybody has left slot sw
        if (g1.dbg) c_wait += __builtin_readcyclecounter() - tw;
#ifndef VFM_MLP_ABL_NODMA
        if (2 * i + e + 2 < 2 * NC / 2 + 2) stage_half(2 * i + e + 2, sw);
#endif
        const unsigned char* sl = ml + (unsigned)sr * SLOT + (unsigned)lane * 16u;
        const unsigned char* s2l = sl + W2OFF;
        const float* bt = tab + (2 * i + e) * 32 + 4 * hi;
        half8 w1r[PF], w2r[PF];
#pragma unroll
        for (int t = 0; t < PF; ++t) {
            w1r[t] = *reinterpret_cast<const half8*>(sl + (unsigned)t * 1024u);
            w2r[t] = *reinterpret_cast<const half8*>(s2l + (unsigned)t * 1024u);
        }
        if constexpr (e == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { hnA[r] = 0.f; hnB[r] = 0.f; }
        }
        float4 bias = *reinterpret_cast<const float4*>(bt), csum = *reinterpret_cast<const float4*>(bt + 1536);
        unsigned pk[8];
        f2 x0, x1, ax0, ax1, z0, z1, t0, t1, p0, p1, e0, e1;   // the group in flight: its two pairs (x0: values 0, 1; x1: values 2, 3)
        // gelu2's arithmetic in six stages; the chunk's four groups take the half's 24 stage slots (two per k-step)
        auto gstep = [&](int ms) __attribute__((always_inline)) {
            const int grp = ms / 6, st = ms % 6;
            if (st == 0) {          // y = acc a + (nb c + b'): epi_tile<EPI_GELU>'s
                x0 = fma2(f2{hc[grp * 4 + 0], hc[grp * 4 + 1]}, A2, fma2(NB2, f2{csum.x, csum.y}, f2{bias.x, bias.y}));
                x1 = fma2(f2{hc[grp * 4 + 2], hc[grp * 4 + 3]}, A2, fma2(NB2, f2{csum.z, csum.w}, f2{bias.z, bias.w}));
                if (grp < 3) {      // the next group's per-channel operands
                    bias = *reinterpret_cast<const float4*>(bt + 8 * (grp + 1));
                    csum = *reinterpret_cast<const float4*>(bt + 1536 + 8 * (grp + 1));
                }
#ifndef VFM_MLP_ABL_NOGELU
            } else if (st == 1) {
                ax0 = f2{fabsf(x0[0]), fabsf(x0[1])};
                ax1 = f2{fabsf(x1[0]), fabsf(x1[1])};
                z0 = ax0 * splat2(0.70710678118654752440f);
                z1 = ax1 * splat2(0.70710678118654752440f);
                const f2 d0 = fma2(splat2(0.3275911f), z0, splat2(1.0f)), d1 = fma2(splat2(0.3275911f), z1, splat2(1.0f));
                t0 = f2{__builtin_amdgcn_rcpf(d0[0]), __builtin_amdgcn_rcpf(d0[1])};
                t1 = f2{__builtin_amdgcn_rcpf(d1[0]), __builtin_amdgcn_rcpf(d1[1])};
            } else if (st == 2) {
                p0 = fma2(splat2(1.061405429f), t0, splat2(-1.453152027f));
                p1 = fma2(splat2(1.061405429f), t1, splat2(-1.453152027f));
                p0 = fma2(p0, t0, splat2(1.421413741f));
                p1 = fma2(p1, t1, splat2(1.421413741f));
                p0 = fma2(p0, t0, splat2(-0.284496736f));
                p1 = fma2(p1, t1, splat2(-0.284496736f));
            } else if (st == 3) {
                p0 = fma2(p0, t0, splat2(0.254829592f));
                p1 = fma2(p1, t1, splat2(0.254829592f));
                const f2 a0 = (z0 * z0) * splat2(-1.44269504088896340736f), a1 = (z1 * z1) * splat2(-1.44269504088896340736f);
                e0 = f2{__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
                e1 = f2{__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
            } else if (st == 4) {
                const f2 r0 = fma2(-(p0 * t0), e0, splat2(1.0f)), r1 = fma2(-(p1 * t1), e1, splat2(1.0f));
                x0 = fma2(ax0 * splat2(0.5f), r0, x0 * splat2(0.5f));
                x1 = fma2(ax1 * splat2(0.5f), r1, x1 * splat2(0.5f));
#endif
            } else if (st == 5) {
                pk[grp * 2 + 0] = pack_f16x2(x0[0], x0[1]);
                pk[grp * 2 + 1] = pack_f16x2(x1[0], x1[1]);
                if (grp & 1) {      // channels 16 s2 .. + 15 of the chunk are complete: to the B-operand layout (the q / P conversion)
                    const int s2 = grp >> 1;
                    auto r0 = __builtin_amdgcn_permlane32_swap(pk[4 * s2 + 0], pk[4 * s2 + 2], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(pk[4 * s2 + 1], pk[4 * s2 + 3], false, false);
                    fout[2 * e + s2] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                }
            }
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 12; ++kk) {
            const int k = 12 * e + kk, pa = 2 * kk, q0 = 2 * kk, q1 = 2 * kk + 1;
            hnA = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1r[pa % PF], *reinterpret_cast<const half8*>(&xf[k]), hnA, 0, 0, 0);
            gstep(2 * kk);
            hnB = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1r[(pa + 1) % PF], *reinterpret_cast<const half8*>(&xf[k]), hnB, 0, 0, 0);
            oacc[q0 % 12] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2r[q0 % PF], *reinterpret_cast<const half8*>(&fin[2 * e + q0 / 12]), oacc[q0 % 12], 0, 0, 0);
            gstep(2 * kk + 1);
            oacc[q1 % 12] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2r[q1 % PF], *reinterpret_cast<const half8*>(&fin[2 * e + q1 / 12]), oacc[q1 % 12], 0, 0, 0);
#ifndef VFM_MLP_ABL_NOLDS
            if (pa + PF < 24) {
                w1r[pa % PF] = *reinterpret_cast<const half8*>(sl + (unsigned)(pa + PF) * 1024u);
                w1r[(pa + 1) % PF] = *reinterpret_cast<const half8*>(sl + (unsigned)(pa + 1 + PF) * 1024u);
                w2r[q0 % PF] = *reinterpret_cast<const half8*>(s2l + (unsigned)(q0 + PF) * 1024u);
                w2r[q1 % PF] = *reinterpret_cast<const half8*>(s2l + (unsigned)(q1 + PF) * 1024u);
            }
#else
            asm volatile("" : "+v"(w1r[pa % PF]), "+v"(w1r[(pa + 1) % PF]), "+v"(w2r[q0 % PF]), "+v"(w2r[q1 % PF]));
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    uint4 fa[4] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)}, fb[4];
    int r = 0;   // slot of half 2 i
    auto nxt = [](int v) { return v + 1 >= 3 ? v - 2 : v + 1; };
#pragma unroll 1
    for (int i = 0; i < NC / 2; i += 2) {
        {   // pair i: reads hA0 / hB0 (its fc1), writes hA1 / hB1 (pair i + 1's), fc2 of pair i - 1 from fa, its own hidden values to fb
            const int s0 = r, s1 = nxt(s0), s2 = nxt(s1);
            half(std::integral_constant<int, 0>{}, i, s0, s2, hA0, hA1, hB1, fa, fb);
            half(std::integral_constant<int, 1>{}, i, s1, s0, hB0, hA1, hB1, fa, fb);
            r = s2;
        }
        {
            const int s0 = r, s1 = nxt(s0), s2 = nxt(s1);
            half(std::integral_constant<int, 0>{}, i + 1, s0, s2, hA1, hA0, hB0, fb, fa);
            half(std::integral_constant<int, 1>{}, i + 1, s1, s0, hB1, hA0, hB0, fb, fa);
            r = s2;
        }
    }
    if (g1.dbg) { tr2 = wall_clock64(); cy2 = __builtin_readcyclecounter(); }
    vit_wait_vmcnt<0>();
    __syncthreads();
    {   // fc2 of the last pair on its own: halves 48 and 49 (chunks 46, 47) in slots r and r + 1; its hidden values are in fa (the loop ran an even number of pairs)
        half8 wr[PF];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const unsigned char* s2l = ml + (unsigned)(e ? nxt(r) : r) * SLOT + W2OFF + (unsigned)lane * 16u;
#pragma unroll
            for (int t = 0; t < PF; ++t) wr[t] = *reinterpret_cast<const half8*>(s2l + (unsigned)t * 1024u);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                oacc[q % 12] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[q % PF], *reinterpret_cast<const half8*>(&fa[2 * e + q / 12]), oacc[q % 12], 0, 0, 0);
                if (q + PF < 24) wr[q % PF] = *reinterpret_cast<const half8*>(s2l + (unsigned)(q + PF) * 1024u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (!live) return;
    // ---- the residual epilogue of fc2 (bias, LayerScale, the stream's read-modify-write, its fp16 copy, the slices' LayerNorm sums)
    const int mtu = __builtin_amdgcn_readfirstlane(mt), qtiles = g2.Tp >> 5;
    const int b = (int)(((unsigned)mtu * g2.qt_magic) >> 20), tq = mtu - b * qtiles, t = tq * 32 + lane31;
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        EpiRegs e;
        epi_load<EPI_RESID>(g2, m, t, hi, j, e);
        epi_tile<EPI_RESID>(g2, oacc[j], m, mtu, b, tq, t, hi, j, e, 0.f, 0.f);
    }
    if (g1.dbg && wave == 0 && lane == 0) {
        g1.dbg[(size_t)blockIdx.x * 8 + 0] = tr0;
        g1.dbg[(size_t)blockIdx.x * 8 + 1] = tr1;
        g1.dbg[(size_t)blockIdx.x * 8 + 2] = tr2;
        g1.dbg[(size_t)blockIdx.x * 8 + 3] = wall_clock64();
        g1.dbg[(size_t)blockIdx.x * 8 + 4] = c_wait;
        g1.dbg[(size_t)blockIdx.x * 8 + 5] = cy2 - cy1;
    }
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) int vfm_vit_launch_mlp_(const void* fc1_args, const void* fc2_args, void* stream) {
    const GemmArgs& g1 = *static_cast<const GemmArgs*>(fc1_args);
    const GemmArgs& g2 = *static_cast<const GemmArgs*>(fc2_args);
    constexpr int lds_ = 144 * 1024 + 2 * 1536 * 4;
    static std::atomic<unsigned long long> attr_set{0ull};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_set >> (dev & 63)) & 1ull)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&vit_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_));
        attr_set |= 1ull << (dev & 63);
    }
    hipLaunchKernelGGL(vit_mlp_kernel, dim3(ceil_div(g1.M / 32, 4)), dim3(256), lds_, (hipStream_t)stream, g1, g2);
    VFM_CHECK_LAUNCH("vit_mlp_kernel");
    return VFM_OK;
}
#endif   // VFM_VIT_PART
