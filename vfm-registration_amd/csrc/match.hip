// match.hip -- descriptor matching on MI355X (gfx950): the whole of
// VoxelHashMap::GetVFMCorrespondences' search (VoxelHashMap.cpp:469-511, 587-600) and
// find_correspondences' nearest neighbours (registration_node.py:482-538).
//
// Two coarse passes feed one exact decision (indices identical to the fp64 oracle either way):
//
// (a) the int8 pass -- the GATED entry points (a caller that keeps only matches above a similarity gate,
//     VoxelHashMap.cpp:501-511), d = 256 ... 768:
//   prep_chunk_kernel       fp32 rows -> 1/|row| (faiss fvec_renorm_L2 order), int8 image of the normalised rows with one
//                           quantisation step per 128-row group, the measured residual norm of every row
//   match_coarse_i8_kernel  int8 MFMA 32x32x32 (exact integer scores), queries resident in VGPRs, map tiles streamed through
//                           an LDS ring by LDS-DMA; per (query, 128-row chunk) the best score; per query a lower bound of its
//                           exact maximum
//   match_select_kernel     per query: chunks whose upper bound reaches the lower bound; queries that cannot reach the gate
//                           are closed
//   match_rescan_kernel     candidate chunks -> candidate rows (exact integer scores of the chunk's rows, v_dot4)
// (b) the fp16 pass -- ungated entry points, d = 128, Euclidean search, duplicate-rich maps:
//   prep_rows_kernel      fp32 rows -> 1/|row| + fp16 copy of the normalised rows in MFMA-fragment ("frag-major") tiles
//   coarse pass           fp16 MFMA 32x32x16, same streaming; sparse row-level records against a running maximum
//                         (match_coarse_pipe_kernel<., true> + match_filter_refine_kernel) or per (query, 128-row chunk)
//                         top-2 records (+ match_select_kernel).  Three shapes:
//                           match_coarse_pipe_kernel  d <= 384: 8 waves, fragment pipeline carried across the step barrier
//                           match_coarse_kernel       d = 512 (ring of 4), and the A/B + ablation base
//                           match_coarse_r_kernel     d = 640 / 768: 4 waves, 192 query registers
// then, common:
//   match_refine_kernel   fp32 scores of crowded candidate lists, rows within the fp32 margin survive
//   match_rescore_kernel  exact fp64 re-decision among the candidates (sequential-k dot of the
//                         fp32-normalised rows, ties -> lowest index)
//   match_exact_kernel    all-pairs fp64 (EXACT mode, and fallback for candidate overflow)
// Euclidean 1-NN (row A6) reuses the coarse pass and select: l2_maxnorm / l2_prep / l2_rescore kernels.
// Compiled with -ffp-contract=off (fp32 sum-of-squares order must match the oracle).
#include "common.h"

#include <type_traits>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

constexpr int TILE_ROWS = 32;     // rows per fragment tile
constexpr int CHUNK_ROWS = 128;   // map rows per partial record (4 tiles)
constexpr int ROW_PAD = 256;      // prepared operands are padded to a multiple of this
constexpr int QBLOCK = 256;       // queries per workgroup of the coarse kernel (8 waves x 32)
// LDS ring depth in tiles: a step consumes 2 tiles; 6 buffers = 2 steps in flight (d <= 384),
// 4 buffers = 1 step in flight when a tile is 32 KiB (d = 512): 160 KiB of LDS per CU.
constexpr int ring_depth(int ksteps) { return ksteps <= 24 ? 6 : 4; }
constexpr int CAND_CAP_MAX = 2048;  // candidate entries a query can hold = min(#chunks, this), a multiple of 64 (cand_cap());
                                    // beyond it the query is decided by the all-pairs kernel.  At C2 (1563 chunks) the cap
                                    // cannot be exceeded: every chunk fits in the list.
constexpr int REFINE_MIN = 3;     // queries with this many candidate entries (or a whole-chunk entry) go through the fp32 refinement
constexpr int REFINE_MIN_I8 = 8;  // the same threshold for the row lists of the int8 pass (match_rescan_kernel)
constexpr int REFINE_KEEP = 64;   // rows a query may keep after the refinement
constexpr int SPARSE_LREC_CAP = 1536;  // records a workgroup of the sparse coarse kernel buffers in LDS (12 KiB)
constexpr float COARSE_OFFSET = 2.0f;   // accumulators start here: every coarse score is a
                                        // positive normal float, so uint order == float order
constexpr float DEFAULT_WINDOW = 2.5e-3f;  // >= 2E, E = proven |coarse - exact| bound
constexpr int I8_OFFSET = 1 << 30;  // int8 coarse pass: accumulators start here (scores positive: uint order == int order)
constexpr int I8_GROUP = 128;       // rows that share one quantisation step (= CHUNK_ROWS: a record never mixes two steps)
static_assert(I8_GROUP == 128, "match_select_kernel and the coarse records assume 128-row groups");

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---------------------------------------------------------------------------------------------
// prepared operand layout: [inv: rows_pad floats][tiles: rows_pad/32 x (d/16 ksteps) x 64 x 16 B]
// unit (tile, s, h, p) = 8 fp16 = row (tile*32+p), k = 16 s + 8 h .. +7  at uint4 index
// tile*(d/16*64) + s*64 + h*32 + p : exactly the register image of one 32x32x16 MFMA operand.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int cand_cap(int64_t map_rows_padded) {
    int64_t c = (map_rows_padded / 128 + 63) / 64 * 64;
    return (int)(c < 64 ? 64 : (c > CAND_CAP_MAX ? CAND_CAP_MAX : c));
}
__host__ __device__ inline int64_t rows_padded(int64_t rows) { return (rows + ROW_PAD - 1) / ROW_PAD * ROW_PAD; }

// sum of squares in the oracle's order: lane l owns the float4 chunks c with c % 64 == l
// (ascending c, ascending element inside a chunk), then an xor butterfly.  d <= 1024.
template <bool STREAM = false>
__device__ __forceinline__ float row_sumsq_wave(const float* __restrict__ row, int d, float4 (&v)[4]) {
    const int lane = lane_id();
    const int nchunks = d >> 2;
    float p = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nchunks) {
            if constexpr (STREAM) {  // read-once data: do not displace the coarse pass' map slice from L2
                const float* pc = row + 4 * c;
                x.x = __builtin_nontemporal_load(pc);
                x.y = __builtin_nontemporal_load(pc + 1);
                x.z = __builtin_nontemporal_load(pc + 2);
                x.w = __builtin_nontemporal_load(pc + 3);
            } else {
                x = reinterpret_cast<const float4*>(row)[c];
            }
            float t;
            t = x.x * x.x; p = p + t;
            t = x.y * x.y; p = p + t;
            t = x.z * x.z; p = p + t;
            t = x.w * x.w; p = p + t;
        }
        v[i] = x;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off);
    return p;
}

__device__ __forceinline__ float inv_norm_from_sumsq(float nr) {
    // faiss: const float inv_nr = 1.0 / sqrtf(nr);  (double divide, rounded to float)
    if (!(nr > 0.0f)) return 0.0f;
    float s = sqrtf(nr);  // __builtin_sqrtf: correctly rounded (HIP default); __fsqrt_rn is the native approximation
    return (float)(1.0 / (double)s);
}

// one workgroup (4 waves) per 32-row tile
// (a second operand -- x2, rows2, ... -- may ride in the same grid: workgroups >= tiles1 prepare it; the scan and the map
// of a registration go out as ONE launch instead of two, 25 us less on the stream the coarse pass waits on)
__global__ __launch_bounds__(256) void prep_rows_kernel(const float* __restrict__ x, int64_t rows, int d,
                                                        float* __restrict__ inv_out,
                                                        uint4* __restrict__ tiles, int tiles1,
                                                        const float* __restrict__ x2, int64_t rows2,
                                                        float* __restrict__ inv_out2, uint4* __restrict__ tiles2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tile = blockIdx.x;
    if (tile >= tiles1) {  // uniform per workgroup
        tile -= tiles1;
        x = x2;
        rows = rows2;
        inv_out = inv_out2;
        tiles = tiles2;
    }
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int ksteps = d >> 4;
    _Float16* img = reinterpret_cast<_Float16*>(smem);
    for (int pr = wave; pr < TILE_ROWS; pr += 4) {
        const int64_t r = (int64_t)tile * TILE_ROWS + pr;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        float inv = 0.0f;
        if (r < rows) {
            float nr = row_sumsq_wave<true>(x + r * (int64_t)d, d, v);
            inv = inv_norm_from_sumsq(nr);
        }
        if (lane == 0) inv_out[r] = inv;
        const int nchunks = d >> 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                const float4 xv = v[i];
                // normalised value exactly as faiss leaves it in fp32, then rounded to fp16 (RNE)
                half4 h;
                h[0] = (_Float16)(xv.x * inv);
                h[1] = (_Float16)(xv.y * inv);
                h[2] = (_Float16)(xv.z * inv);
                h[3] = (_Float16)(xv.w * inv);
                const int s = c >> 2, hh = (c >> 1) & 1, sub = c & 1;
                const int unit = (s * 2 + hh) * 32 + pr;
                *reinterpret_cast<half4*>(img + unit * 8 + sub * 4) = h;
            }
        }
    }
    __syncthreads();
    const int units = ksteps * 64;
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

// ---------------------------------------------------------------------------------------------
// int8 image of the rows (d = 256, 384) for the int8 coarse pass, plus (F16) the fp16 image above.
// One workgroup (8 waves) per GROUP of 128 rows = one record chunk of the coarse pass.  The rows of a group share one
// quantisation step  s = max|v| / 127  over the group's fp32-normalised elements v (no clipping, no tuning constant):
//     q_k = rint(v_k / s) in [-127, 127],   e = v - s q  (measured in fp32, not assumed),   E = |e|_2 rounded up.
// The integer score S = q_a . q_b of the MFMA is exact, so for rows a (step s_a) and b (step s_b)
//     | v_a . v_b - s_a s_b S |  =  | (s_a q_a) . e_b + e_a . v_b |  <=  (|v_a| + E_a) E_b + E_a |v_b|     (Cauchy-Schwarz)
// with |v| <= 1 + 2^-13 for fp32-normalised rows: match_select_kernel turns this into per-(query, chunk) bounds.
// 16 waves x 8 rows: the group's rows stay in registers between phase 1 (1/|row|, group maximum) and phase 2 (quantise).
// Layout: int8 fragment tiles of the 32x32x32 MFMA: unit (tile, s, h, p) = 16 int8 = row tile*32+p, k = 32 s + 16 h .. +15,
// at uint4 index tile*(d/32*64) + s*64 + h*32 + p.
// ---------------------------------------------------------------------------------------------
struct PrepOut {
    float* inv;       // [rows_pad]
    uint4* tiles;     // fp16 fragment tiles
    float* err;       // [rows_pad] E per row
    float* gstep;     // [rows_pad / 128] quantisation step of the group
    float* gerr;      // [rows_pad / 128] maximum E of the group
    uint4* tiles8;    // int8 fragment tiles
};
template <bool F16, int NC = 2>
__global__ __launch_bounds__(1024) void prep_chunk_kernel(const float* __restrict__ x, int64_t rows, int d, PrepOut o, int groups1,
                                                          const float* __restrict__ x2, int64_t rows2, PrepOut o2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned amax_bits, emax_bits;
    constexpr int RPW = I8_GROUP / 16;  // rows per wave: 16 waves x 8 rows, all of them in registers between the two phases
    int grp = blockIdx.x;
    if (grp >= groups1) {  // uniform per workgroup: the second operand rides in the same grid
        grp -= groups1;
        x = x2;
        rows = rows2;
        o = o2;
    }
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int nchunks = d >> 2;  // float4 chunks per row (<= 64 NC): lane l owns chunks l, l + 64 (, l + 128)
    unsigned char* img8 = smem;                                                  // [4 tiles][d/32 * 64 units][16]
    _Float16* img16 = reinterpret_cast<_Float16*>(smem + (size_t)I8_GROUP * d);  // F16: [4 tiles][d/16 * 64 units][8]
    if (threadIdx.x == 0) {
        amax_bits = 0u;
        emax_bits = 0u;
    }
    // phase 1: the rows (read once), 1/|row| in the oracle's order, the group's largest normalised magnitude
    float4 v[RPW][NC];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int64_t r = (int64_t)grp * I8_GROUP + wave * RPW + j;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && c < nchunks) {
                const float* pc = x + r * (int64_t)d + 4 * c;
                t.x = __builtin_nontemporal_load(pc);
                t.y = __builtin_nontemporal_load(pc + 1);
                t.z = __builtin_nontemporal_load(pc + 2);
                t.w = __builtin_nontemporal_load(pc + 3);
            }
            v[j][i] = t;
        }
    }
    // Eight per-row sums per lane -> one per lane: a reduce-scatter over the xor-32 / 16 / 8 levels (the lane keeps half of
    // its rows at every level and adds the partner's partial of those rows), then the xor-4 / 2 / 1 levels on the single
    // value.  Every addition pairs the same two partials as row_sumsq_wave's butterfly (which computes each of them in both
    // lanes), so the sum is bit-identical to the oracle's; 10 shuffles instead of 48.  Afterwards lane l holds row l >> 3.
    auto scatter8 = [&](float (&p)[RPW]) __attribute__((always_inline)) {
        const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
        float q4[4], q2[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) q4[j] = (b5 ? p[j + 4] : p[j]) + __shfl_xor(b5 ? p[j] : p[j + 4], 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) q2[j] = (b4 ? q4[j + 2] : q4[j]) + __shfl_xor(b4 ? q4[j] : q4[j + 2], 16);
        float q1 = (b3 ? q2[1] : q2[0]) + __shfl_xor(b3 ? q2[0] : q2[1], 8);
        q1 = q1 + __shfl_xor(q1, 4);
        q1 = q1 + __shfl_xor(q1, 2);
        q1 = q1 + __shfl_xor(q1, 1);
        return q1;
    };
    float part[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {  // lane-sequential over its chunks and elements, as row_sumsq_wave
        float p = 0.0f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (lane + 64 * i < nchunks) {
                float t;
                t = v[j][i].x * v[j][i].x; p = p + t;
                t = v[j][i].y * v[j][i].y; p = p + t;
                t = v[j][i].z * v[j][i].z; p = p + t;
                t = v[j][i].w * v[j][i].w; p = p + t;
            }
        }
        part[j] = p;
    }
    const float my_inv = inv_norm_from_sumsq(scatter8(part));  // of row lane >> 3: eight rows in one evaluation
    if ((lane & 7) == 0) o.inv[(int64_t)grp * I8_GROUP + wave * RPW + (lane >> 3)] = my_inv;
    float lmax = 0.0f;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const float iv = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_inv), 8 * j));
#pragma unroll
        for (int i = 0; i < NC; ++i) {  // normalised values exactly as faiss leaves them in fp32 (zero in the unused slots)
            v[j][i].x = v[j][i].x * iv;
            v[j][i].y = v[j][i].y * iv;
            v[j][i].z = v[j][i].z * iv;
            v[j][i].w = v[j][i].w * iv;
            lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(v[j][i].x), fabsf(v[j][i].y)), fmaxf(fabsf(v[j][i].z), fabsf(v[j][i].w))));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    __syncthreads();  // amax_bits / emax_bits initialised
    if (lane == 0 && lmax > 0.0f) atomicMax(&amax_bits, __float_as_uint(lmax));  // finite or +Inf: uint order == float order
    __syncthreads();
    const float amax = __uint_as_float(amax_bits);
    const bool usable = amax > 0.0f && amax < 3.0e38f;
    const float qstep = usable ? amax / 127.0f : 1.0f;
    const float inv_qstep = usable ? 127.0f / amax : 0.0f;
    // phase 2: quantise from the registers
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int pr = wave * RPW + j;
        const int t = pr >> 5, p = pr & 31;
        float e2 = 0.0f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunks) {
                const float nv[4] = {v[j][i].x, v[j][i].y, v[j][i].z, v[j][i].w};
                int qi[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float qf = rintf(nv[e] * inv_qstep);
                    qf = fminf(fmaxf(qf, -127.0f), 127.0f);                // (a NaN becomes -127: any integer is valid,
                    const float res = __builtin_fmaf(-qstep, qf, nv[e]);   //  the residual is measured: it turns E into Inf)
                    e2 = __builtin_fmaf(res, res, e2);
                    qi[e] = (int)qf;
                }
                // low bytes of the four integers: two byte permutes and an or
                const unsigned packed = __builtin_amdgcn_perm((unsigned)qi[1], (unsigned)qi[0], 0x0c0c0400u) |
                                        __builtin_amdgcn_perm((unsigned)qi[3], (unsigned)qi[2], 0x04000c0cu);
                *reinterpret_cast<unsigned*>(img8 + (size_t)t * (d * 32) + (((c >> 3) * 2 + ((c >> 2) & 1)) * 32 + p) * 16 + (c & 3) * 4) = packed;
                if constexpr (F16) {
                    half4 h;
                    h[0] = (_Float16)nv[0];
                    h[1] = (_Float16)nv[1];
                    h[2] = (_Float16)nv[2];
                    h[3] = (_Float16)nv[3];
                    const int s = c >> 2, hh = (c >> 1) & 1, sub = c & 1;
                    *reinterpret_cast<half4*>(img16 + (size_t)t * (d * 32) + ((s * 2 + hh) * 32 + p) * 8 + sub * 4) = h;
                }
            }
        }
        part[j] = e2;
    }
    {
        // |e|_2 of row lane >> 3, rounded up: the fp32 sum of d non-negative terms is within (d + 8) 2^-24 of exact, sqrtf
        // within 2^-24
        float en = sqrtf(scatter8(part)) * 1.000244140625f + 1.0e-30f;
        if (!(en == en)) en = __builtin_inff();
        const int64_t r = (int64_t)grp * I8_GROUP + wave * RPW + (lane >> 3);
        if (r >= rows) en = 0.0f;
        if ((lane & 7) == 0) {
            o.err[r] = en;
            if (en > 0.0f) atomicMax(&emax_bits, __float_as_uint(en));
        }
    }
    __syncthreads();
    {
        const int u8n = (d >> 5) * 64 * 4;  // uint4 units of the group's four int8 tiles
        uint4* dst = o.tiles8 + (int64_t)grp * u8n;
        const uint4* src = reinterpret_cast<const uint4*>(img8);
        for (int u = threadIdx.x; u < u8n; u += 1024) {
            const uint4 tq = src[u];
            unsigned* po = reinterpret_cast<unsigned*>(dst + u);
            __builtin_nontemporal_store(tq.x, po);
            __builtin_nontemporal_store(tq.y, po + 1);
            __builtin_nontemporal_store(tq.z, po + 2);
            __builtin_nontemporal_store(tq.w, po + 3);
        }
    }
    if constexpr (F16) {
        const int u16n = (d >> 4) * 64 * 4;
        uint4* dst = o.tiles + (int64_t)grp * u16n;
        const uint4* src = reinterpret_cast<const uint4*>(img16);
        for (int u = threadIdx.x; u < u16n; u += 1024) {
            const uint4 tq = src[u];
            unsigned* po = reinterpret_cast<unsigned*>(dst + u);
            __builtin_nontemporal_store(tq.x, po);
            __builtin_nontemporal_store(tq.y, po + 1);
            __builtin_nontemporal_store(tq.z, po + 2);
            __builtin_nontemporal_store(tq.w, po + 3);
        }
    }
    if (threadIdx.x == 0) {
        o.gstep[grp] = qstep;
        o.gerr[grp] = __uint_as_float(emax_bits);
    }
}

// in-place renorm (vfm_l2norm_rows_f32): one wave per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ x, int64_t rows, int d,
                                                          float* __restrict__ inv_out) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float4 v[4];
    float* row = x + r * (int64_t)d;
    float nr = row_sumsq_wave(row, d, v);
    float inv = inv_norm_from_sumsq(nr);
    if (lane_id() == 0 && inv_out) inv_out[r] = inv;
    if (!(nr > 0.0f)) return;
    const int nchunks = d >> 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane_id() + 64 * i;
        if (c < nchunks) {
            float4 xv = v[i];
            xv.x = xv.x * inv; xv.y = xv.y * inv; xv.z = xv.z * inv; xv.w = xv.w * inv;
            reinterpret_cast<float4*>(row)[c] = xv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// coarse pass
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else static_assert(N < 0, "unsupported vmcnt");
}

// LDS-DMA, 16 B per lane: LDS destination = wave-uniform byte address in M0 + lane * 16.
// Issued from inline asm so that hipcc neither counts it nor drains it (it would place an
// s_waitcnt vmcnt(0) in front of the next ds_read); completion is tracked by wait_vmcnt<N>().
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
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

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
    return max(min(a, b), min(max(a, b), c));
}

// per-row / per-group quantisation data of the two operands of an int8 pass (qerr == NULL: fp16 records)
struct I8Bounds {
    const float* qerr;   // [npad] E of every query row
    const float* qstep;  // [npad / 128] step of the query's group
    const float* bstep;  // [nchunks] step of the map chunk
    const float* berr;   // [nchunks] maximum E of the map chunk
    int top2;            // records: 0 = best score per (query, chunk), 1 = packed top-2 with the best row's index
};
// float <-> unsigned key with the same order (0 = below every float: the memset value of "nothing published")
__device__ __forceinline__ unsigned float_key(float f) {
    const int k = __float_as_int(f);
    return (unsigned)(k >= 0 ? k : k ^ 0x7FFFFFFF) ^ 0x80000000u;
}
__device__ __forceinline__ float key_float(unsigned u) {
    if (u == 0u) return -__builtin_inff();
    const int k = (int)(u ^ 0x80000000u);
    return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF);
}

struct CoarseArgs {
    const uint4* Qh;     // query fragment tiles
    const uint4* Bh;     // map fragment tiles
    uint2* partials;     // [nchunks][npad]
    int nq_tiles;        // valid 32-query tiles
    int nchunks;         // map chunks (128 rows)
    long long m_valid;   // real map rows
    int npad;            // padded query count (row stride of partials)
    int nqb;             // query blocks (256 queries)
    int nslices;         // map slices
    unsigned* qmax;      // [npad] running coarse maximum per query (value bits only), zeroed per search
    int first_pad_chunk; // chunks >= this contain zero-padded map rows: excluded from qmax
    const float* row_bias;  // [padded map rows] added to the accumulator start of that row, or NULL
                            // (Euclidean search for d > 510: -|b~|^2 / 2; match_coarse_r_kernel only)
    // sparse row-level records (match_coarse_pipe_kernel<., true>): every (query, map row) whose coarse score is
    // within `window` of the query's running maximum at that time -- a superset of the rows within `window` of the
    // final maximum, which is all the exact decision needs
    const float* qinv;   // [npad] 1/|query row| (0 for zero rows: they record nothing, match_rescore_kernel decides them)
    // seed units (sparse path): the first seed_parts * seed_chunks chunks of the map are taken by short workgroups at the
    // head of the grid, seed_parts per query block, so that every later unit of a query block starts from a published
    // maximum (a fresh running maximum breaks records at rate ~1/k per row; with 55 slices, 15 % of the units used to
    // start unseeded and the record-breaking phase cost the kernel 5 %)
    int seed_parts, seed_chunks, nseed_pad;
    unsigned* rec_cnt;   // [npad] records appended per query (may exceed rcap: overflow)
    uint2* rec;          // [npad][rcap] (map row, score bits)
    int rcap;
    float window;
    I8Bounds ib;         // int8 pass: qmax receives float_key(lower bound of the query's exact maximum) instead of score bits
};

// XCD-aware unit mapping shared by the coarse kernels: workgroup b runs on XCD b % 8 (observed, speed
// only); every XCD gets one contiguous range of (slice-major) units so that co-resident workgroups stream
// the same map slice through that XCD's L2.  Unit = (query block qb, map slice); tiles [4 c0, 4 c1).
struct CoarseUnit {
    int qb, c0, ntiles;
};
__device__ __forceinline__ CoarseUnit coarse_unit(const CoarseArgs& a) {
    const int total = a.nqb * a.nslices;
    int bid = blockIdx.x;
    const int seed_total = a.seed_parts * a.seed_chunks;  // chunks [0, seed_total) belong to the seed units
    if (bid < a.nseed_pad) {  // nseed_pad is a multiple of 8: the XCD phase of the remaining grid is unchanged
        CoarseUnit u;
        u.qb = bid / a.seed_parts;
        u.c0 = (bid - u.qb * a.seed_parts) * a.seed_chunks;
        u.ntiles = (u.qb < a.nqb) ? a.seed_chunks * 4 : 0;  // padding workgroups of the seed round: nothing to do
        return u;
    }
    bid -= a.nseed_pad;
    const int xcd = bid & 7, within = bid >> 3;
    const int qn = total >> 3, rn = total & 7;
    const int unit = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + within;
    const int slice = unit / a.nqb;
    CoarseUnit u;
    u.qb = unit - slice * a.nqb;
    const int rest = a.nchunks - seed_total;
    u.c0 = seed_total + (int)(((long long)slice * rest) / a.nslices);
    const int c1 = seed_total + (int)(((long long)(slice + 1) * rest) / a.nslices);
    u.ntiles = (c1 - u.c0) * 4;
    return u;
}

// One accumulator element into the running top-2 of its chunk: 3 VALU ops, branch-free.  code = 16 * tile
// in chunk + accumulator register; zero-padded map rows (score exactly 2.0) are NOT masked here:
// match_select_kernel ignores padded chunks for the maximum and rescans them exactly.
__device__ __forceinline__ void coarse_fold_bits(unsigned& s1, unsigned& s2, unsigned bits, int code) {
    const unsigned pk = (bits & 0xFFFFFFC0u) | (unsigned)(63 - code);
    s2 = umed3(s1, s2, pk);
    s1 = max(s1, pk);
}
__device__ __forceinline__ void coarse_fold(unsigned& s1, unsigned& s2, float v, int code) { coarse_fold_bits(s1, s2, __float_as_uint(v), code); }
__device__ __forceinline__ void coarse_fold(unsigned& s1, unsigned& s2, int v, int code) { coarse_fold_bits(s1, s2, (unsigned)v, code); }
__device__ __forceinline__ unsigned score_bits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ unsigned score_bits(int v) { return (unsigned)v; }

// End of a 128-row chunk: merge the two half-waves and emit the chunk's top-2 for the 32 queries of tile
// qt (chunk < 0: the dummy fold of the very first step, nothing is stored); resets the running pair.
__device__ __forceinline__ unsigned coarse_emit_chunk(const CoarseArgs& a, unsigned& s1, unsigned& s2, unsigned& runmax,
                                                      int qt, int chunk) {
    const int lane = lane_id(), hi = lane >> 5;
    const unsigned o1 = __shfl_xor(s1, 32), o2 = __shfl_xor(s2, 32);
    const bool own = (s1 > o1) || (s1 == o1 && hi == 0);
    const unsigned w1 = own ? s1 : o1;
    const int wh = own ? hi : (1 - hi);
    const unsigned w2 = max(max(s2, o2), min(s1, o1));
    const int code = 63 - (int)(w1 & 63u);
    const int li = (code >> 4) * 32 + (code & 3) + 8 * ((code & 15) >> 2) + 4 * wh;  // row inside the chunk
    if (lane < 32 && qt < a.nq_tiles && chunk >= 0) {
        a.partials[(size_t)chunk * a.npad + (size_t)qt * 32 + lane] = make_uint2((w1 & ~127u) | (unsigned)li, w2);
        if (chunk < a.first_pad_chunk) runmax = max(runmax, w1 & ~127u);
    }
    s1 = 0u;
    s2 = 0u;
    return w1 & ~127u;  // the chunk's best score, low 7 bits dropped (all lanes)
}

// int8 pass: the chunk's BEST VALUE only -- one VALU op per accumulator element instead of three (with the packed top-2 the
// fold had become the kernel's limiter: the int8 MFMA halves the matrix time per element -- 5 VALU ops per MFMA, matrix pipe
// 56 % busy) and a 4-byte record.  Which rows of a candidate chunk matter is found by match_refine_kernel's int8 rescan.
__device__ __forceinline__ unsigned coarse_emit_chunk_best(const CoarseArgs& a, unsigned& s1, int qt, int chunk) {
    const int lane = lane_id();
    const unsigned w1 = max(s1, (unsigned)__shfl_xor(s1, 32));
    // layout [query tile][chunk][32]: a wave's records of consecutive chunks are consecutive 128-byte lines, and
    // match_select_kernel reads each query tile as one contiguous stream
    if (lane < 32 && qt < a.nq_tiles && chunk >= 0)
        reinterpret_cast<unsigned*>(a.partials)[((size_t)qt * a.nchunks + (size_t)chunk) * 32 + lane] = w1;
    s1 = 0u;
    return w1;
}

// QSETS = 32-query sets resident per wave: 1 -> 8 waves (2 per SIMD), 2 -> 4 waves (1 per SIMD,
// every LDS fragment feeds two MFMAs: half the LDS read traffic / energy per flop).
template <int KSTEPS, int QSETS>
__global__ __launch_bounds__(512 / QSETS, (QSETS == 1) ? 2 : 1) void match_coarse_kernel(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVES = 8 / QSETS;
    constexpr int TILE_U4 = KSTEPS * 64;
    constexpr int TILE_BYTES = TILE_U4 * 16;
    constexpr int PASSES = KSTEPS / NWAVES;  // 1 KiB pieces per wave per tile
    constexpr int NBUF = ring_depth(KSTEPS);
    constexpr int AHEAD = NBUF / 2 - 1;  // steps of prefetch distance
    static_assert(KSTEPS % 8 == 0, "d must be a multiple of 128");

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const CoarseUnit cu = coarse_unit(a);
    const int qb = cu.qb, c0 = cu.c0, ntiles = cu.ntiles;
    const int t0 = c0 * 4;

    const int qt0 = qb * 8 + wave * QSETS;  // first 32-query tile of this wave

    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS unsigned char*)smem;
    auto stage = [&](int it_s) {
        const uint4* src = a.Bh + (size_t)(t0 + it_s) * TILE_U4;
        const unsigned dst = lds_base + (unsigned)((it_s % NBUF) * TILE_BYTES);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int piece = p * NWAVES + wave;
            glds16(src + piece * 64 + lane, __builtin_amdgcn_readfirstlane(dst + (unsigned)piece * 1024u));
        }
    };

    // query fragments stay in registers for the whole slice
    half8 qf[QSETS][KSTEPS];
#pragma unroll
    for (int j = 0; j < QSETS; ++j) {
        const int qt = qt0 + j;
        const uint4* qsrc = a.Qh + (size_t)(qt < a.nq_tiles ? qt : 0) * TILE_U4 + lane;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            uint4 v = qsrc[s * 64];
            qf[j][s] = *reinterpret_cast<half8*>(&v);
        }
    }
#pragma unroll
    for (int i = 0; i < 2 * AHEAD; ++i)
        if (i < ntiles) stage(i);  // ntiles is a multiple of 4 (whole chunks)

    unsigned s1[QSETS], s2[QSETS], runmax[QSETS];
#pragma unroll
    for (int j = 0; j < QSETS; ++j) s1[j] = s2[j] = runmax[j] = 0u;

    // epilogue of one finished 32 x 32 accumulator tile: fold into the chunk's running top-2 (branch-free
    // so that the scheduler can issue it inside the next step's MFMA cluster); tile 3 closes the chunk
    auto fold = [&](const floatx16& acc, int it, auto TTc, auto Jc) {
        constexpr int TT = decltype(TTc)::value;
        constexpr int J = decltype(Jc)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) coarse_fold(s1[J], s2[J], acc[r], TT * 16 + r);
        if constexpr (TT == 3) coarse_emit_chunk(a, s1[J], s2[J], runmax[J], qt0 + J, it >= 0 ? c0 + (it >> 2) : -1);
    };

    // one step = 2 map tiles (64 rows): 2 * QSETS independent accumulator chains per wave, one
    // barrier.  The top-2 fold of step i-1 (VALU) is issued inside the MFMA cluster of step i
    // (matrix pipe), so the two pipes overlap within a wave instead of alternating in lockstep.
    floatx16 prev[QSETS][2];
#pragma unroll
    for (int j = 0; j < QSETS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            prev[j][0][r] = 0.f;
            prev[j][1][r] = 0.f;
        }
    auto do_step = [&](int it, auto Hc) {
        constexpr int H = decltype(Hc)::value;  // which half of the 4-tile chunk
        // tiles it, it+1 must have landed; the (AHEAD-1) newer steps may stay in flight
        if (AHEAD >= 2 && it + 2 < ntiles) wait_vmcnt<2 * PASSES>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifndef VFM_ABLATE_DMA
        if (it + 2 * AHEAD < ntiles) {
            stage(it + 2 * AHEAD);
            stage(it + 2 * AHEAD + 1);
        }
#endif
        const uint4* buf0 = reinterpret_cast<const uint4*>(smem + (it % NBUF) * TILE_BYTES) + lane;
        const uint4* buf1 = reinterpret_cast<const uint4*>(smem + ((it + 1) % NBUF) * TILE_BYTES) + lane;
        floatx16 acc[QSETS][2];
#pragma unroll
        for (int j = 0; j < QSETS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[j][0][r] = COARSE_OFFSET;
                acc[j][1][r] = COARSE_OFFSET;
            }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
#ifdef VFM_ABLATE_LDS
            uint4 v0 = *reinterpret_cast<const uint4*>(&qf[0][(s + 1) % KSTEPS]), v1 = *reinterpret_cast<const uint4*>(&qf[0][(s + 2) % KSTEPS]);
            (void)buf0; (void)buf1;
#else
            uint4 v0 = buf0[s * 64], v1 = buf1[s * 64];
#endif
#pragma unroll
            for (int j = 0; j < QSETS; ++j) {
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&v0), qf[j][s], acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&v1), qf[j][s], acc[j][1], 0, 0, 0);
            }
        }
#ifdef VFM_ABLATE_FOLD
#pragma unroll
        for (int j = 0; j < QSETS; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(acc[j][0]), "v"(acc[j][1]));
#endif
        }
        if (it >= 0) return;
#endif
        // previous step = the other half (of this chunk for H == 1, of the previous chunk for H == 0);
        // at it == 0 this folds the zero-initialised dummies (tiny packed values, store suppressed)
        fold(prev[0][0], it - 2, std::integral_constant<int, 2 * (1 - H)>{}, std::integral_constant<int, 0>{});
        fold(prev[0][1], it - 1, std::integral_constant<int, 2 * (1 - H) + 1>{}, std::integral_constant<int, 0>{});
        if constexpr (QSETS == 2) {
            fold(prev[1][0], it - 2, std::integral_constant<int, 2 * (1 - H)>{}, std::integral_constant<int, 1>{});
            fold(prev[1][1], it - 1, std::integral_constant<int, 2 * (1 - H) + 1>{}, std::integral_constant<int, 1>{});
        }
#pragma unroll
        for (int j = 0; j < QSETS; ++j) {
            prev[j][0] = acc[j][0];
            prev[j][1] = acc[j][1];
        }
    };

    for (int it = 0; it < ntiles; it += 4) {
        do_step(it, std::integral_constant<int, 0>{});
        do_step(it + 2, std::integral_constant<int, 1>{});
    }
    fold(prev[0][0], ntiles - 2, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
    fold(prev[0][1], ntiles - 1, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
    if constexpr (QSETS == 2) {
        fold(prev[1][0], ntiles - 2, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
        fold(prev[1][1], ntiles - 1, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
    }
    // one atomic per (query, slice): the per-query coarse maximum match_select_kernel thresholds on
#pragma unroll
    for (int j = 0; j < QSETS; ++j)
        if (lane < 32 && qt0 + j < a.nq_tiles) atomicMax(a.qmax + (size_t)(qt0 + j) * 32 + lane, runmax[j]);
}

// ---------------------------------------------------------------------------------------------
// coarse pass, pipelined across the step barrier (default for d <= 384)
//
// Same tiles, ring, fold and results as match_coarse_kernel<KSTEPS, 1>; what changes is the order
// of work around the one barrier per step.  PMC showed the old kernel's waves 34 % of their time
// in s_waitcnt/s_barrier with the LDS only 34 % busy: after each barrier BOTH waves of a SIMD ran
// the serial restart (ring arithmetic, DMA issue, first ds_read latency) while the matrix pipe
// idled.  Here
//   * map tiles land one step early (s_waitcnt vmcnt(0) at the top of step i covers the tiles of
//     step i+1), so the last PF slots of step i read the first PF fragments of step i+1: the
//     MFMAs after the barrier start from registers;
//   * fragment reads run PF k-steps ahead of their MFMAs, the deferred top-2 fold of step i-1 is
//     spread over the slots, the LDS-DMA of step i+2's tiles is issued one 1 KiB piece per slot from
//     slot 1 on (a burst of all 48 pieces of the workgroup right after the barrier measured 1.3 % slower);
//     __builtin_amdgcn_sched_barrier(0) keeps the slots apart;
//   * ring offsets are carried incrementally (no division in the loop).
// Ring of 6 tiles: step i computes on (i, i+1), prefetches from (i+2, i+3), DMA fills (i+4, i+5)
// = the slots of (i-2, i-1), whose last read retired before the barrier of step i.
// ---------------------------------------------------------------------------------------------
//
// SPARSE = true (inner-product search, the default): instead of one top-2 record per (query, 128-row chunk) -- 253 MB
// at C2, swept again by match_select_kernel, and ambiguous whenever two rows of a chunk are both inside the window --
// the epilogue keeps ONE running maximum per lane (1 VALU op per accumulator element instead of 3) and, once per step,
// tests the step's maximum against `running maximum - window`; only then (rare: a lane's query meets a near-best row)
// are the 32 accumulators of the previous step compared one by one and the hits appended to the query's record list
// (atomic slot counter).  The running maximum starts from the maxima earlier workgroups published for the query
// (a.qmax), so only the first units of a query see the record-breaking phase of a fresh maximum.
template <int KSTEPS, bool SPARSE>
__global__ __launch_bounds__(512, 2) void match_coarse_pipe_kernel(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVES = 8;
    constexpr int TILE_U4 = KSTEPS * 64;
    constexpr int TILE_BYTES = TILE_U4 * 16;
    constexpr int PIECES = 2 * KSTEPS / NWAVES;  // 1 KiB pieces per wave per PAIR of tiles (one step)
    constexpr int NBUF = 6;
    constexpr int PF = 4;
    static_assert((2 * KSTEPS) % NWAVES == 0 && KSTEPS <= 24 && PIECES < KSTEPS, "a pair of tiles must split evenly over the waves");
    static_assert(KSTEPS % PF == 0 && KSTEPS >= 2 * PF, "fragment ring must align across steps");
    using frag_t = half8;
    using acc_t = floatx16;
    using accel_t = float;

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const CoarseUnit cu = coarse_unit(a);
    const int qb = cu.qb, c0 = cu.c0, ntiles = cu.ntiles;
    if (ntiles == 0) return;  // uniform: padding workgroup of the seed round
    const int qt = qb * NWAVES + wave;  // this wave's 32-query tile

    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS unsigned char*)smem;
    // this wave's pieces of one tile: global source of piece p = src + p * NWAVES * 64, LDS p * NWAVES KiB on
    const uint4* gsrc = a.Bh + (size_t)c0 * 4 * TILE_U4 + wave * 64 + lane;
    const unsigned ldst0 = lds_base + (unsigned)wave * 1024u;
    auto stage_pair = [&](const uint4* src, unsigned ring_byte) {  // two consecutive tiles: contiguous in memory and in the ring
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            glds16(src + p * NWAVES * 64, __builtin_amdgcn_readfirstlane(ldst0 + ring_byte + (unsigned)(p * NWAVES) * 1024u));
    };

    frag_t qf[KSTEPS];
    {
        const uint4* qsrc = a.Qh + (size_t)(qt < a.nq_tiles ? qt : 0) * TILE_U4 + lane;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            uint4 v = qsrc[s * 64];
            qf[s] = *reinterpret_cast<frag_t*>(&v);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)  // ntiles >= 4 (whole chunks)
        stage_pair(gsrc + (size_t)i * 2 * TILE_U4, (unsigned)(i * 2 * TILE_BYTES));
    const uint4* gnext = gsrc + (size_t)4 * TILE_U4;  // next tile to stage

    unsigned s1 = 0u, s2 = 0u, runmax = 0u;
    constexpr int LREC_CAP = SPARSE_LREC_CAP;
    uint2* lrec = reinterpret_cast<uint2*>(smem + NBUF * TILE_BYTES);        // [LREC_CAP] (query in block << 24 | row, score bits)
    unsigned* lrec_count = reinterpret_cast<unsigned*>(lrec + LREC_CAP);
    if constexpr (SPARSE) {
        if (qt < a.nq_tiles) {
            // published by earlier units (any stale value is valid).  Device-scope load: the publishing atomicMax is
            // performed at device scope, but a plain load could be served from this XCD's own (non-coherent) L2 and
            // keep returning the zero it cached at the start of the launch.
            runmax = __hip_atomic_load(a.qmax + (size_t)qt * 32 + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // a zero query row scores exactly 2.0 against every map row: it would record all of them.  Its answer is
            // fixed (index 0, score 0): park its maximum at the largest float so that nothing passes the threshold.
            if (a.qinv[(size_t)qt * 32 + (lane & 31)] == 0.0f) runmax = 0x7F7FFFFFu;
        }
        if (threadIdx.x == 0) *lrec_count = 0u;  // visible after the first barrier below
    }
    auto fold_tail = [&](int it) {
        if constexpr (!SPARSE) coarse_emit_chunk(a, s1, s2, runmax, qt, it >= 0 ? c0 + (it >> 2) : -1);
    };
    auto fold_one = [&](accel_t v, int code) {
        if constexpr (SPARSE) {
            s1 = max(s1, score_bits(v));  // s1 = maximum of the step being folded
        } else {
            coarse_fold(s1, s2, v, code);
        }
    };
    // SPARSE: end of the fold of tiles (t0, t0 + 1) of this unit, whose accumulators are still in p0 / p1
    auto step_tail = [&](int t0, const acc_t& p0, const acc_t& p1) __attribute__((always_inline)) {
        const long long row0 = ((long long)c0 * 4 + t0) * TILE_ROWS;
        const int half4 = 4 * (lane >> 5);
        const bool pad = row0 + 2 * TILE_ROWS > a.m_valid;
        if (pad) {  // wave-uniform, last tiles of the map only: zero-padded rows score exactly
            s1 = 0u;                                // 2.0 and must neither raise the maximum nor be recorded
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long ra = row0 + (r & 3) + 8 * (r >> 2) + half4;
                if (ra < a.m_valid) s1 = max(s1, score_bits(p0[r]));
                if (ra + TILE_ROWS < a.m_valid) s1 = max(s1, score_bits(p1[r]));
            }
        }
        runmax = max(runmax, s1);
        {   // the other half-wave folds the other 16 rows per tile: v_permlane32_swap (VALU; a ds_bpermute would make the
            // wave wait on lgkmcnt(0), i.e. on the fragment prefetches of the next step)
            const auto sw = __builtin_amdgcn_permlane32_swap(runmax, runmax, false, false);
            runmax = max((unsigned)sw[0], (unsigned)sw[1]);
        }
        const unsigned thr = __float_as_uint(__uint_as_float(runmax) - a.window);
        if (s1 >= thr && qt < a.nq_tiles) {  // rare
            // Hits go to a workgroup buffer in LDS (slot from an LDS atomic: waits on lgkmcnt only).  A returning
            // GLOBAL atomic here would make the wave wait on vmcnt(0), i.e. on every LDS-DMA tile in flight: measured
            // +9 % kernel time.  The buffer is flushed to the per-query lists after the last step.
            // The 32 accumulators are searched in groups of four (3 max + 1 test per group, element tests only inside a
            // group that holds a hit): an entry costs ~70 instead of ~250 instructions -- the whole wave pays for it.
            const unsigned ql = (unsigned)(wave * 32 + (lane & 31));
            auto emit = [&](unsigned x, long long row) __attribute__((always_inline)) {
                const unsigned slot = atomicAdd(lrec_count, 1u);
                if (slot < (unsigned)LREC_CAP) {
                    lrec[slot] = make_uint2((ql << 24) | (unsigned)row, x);
                } else {  // buffer full (a fresh maximum meeting a duplicate-rich map): straight to the list
                    const size_t qi = (size_t)qt * 32 + (lane & 31);
                    const unsigned gs = atomicAdd(a.rec_cnt + qi, 1u);
                    if (gs < (unsigned)a.rcap) a.rec[qi * (size_t)a.rcap + gs] = make_uint2((unsigned)row, x);
                }
            };
#pragma unroll
            for (int g4 = 0; g4 < 8; ++g4) {
                unsigned x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = score_bits(g4 < 4 ? p0[4 * g4 + e] : p1[4 * (g4 - 4) + e]);
                if (max(max(x[0], x[1]), max(x[2], x[3])) >= thr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // accumulator register r = 4 (g4 & 3) + e of tile (g4 >> 2): row (r & 3) + 8 (r >> 2) + 4 * half
                        const long long row = row0 + (g4 >> 2) * TILE_ROWS + e + 8 * (g4 & 3) + half4;
                        if (x[e] >= thr && (!pad || row < a.m_valid)) emit(x[e], row);
                    }
                }
            }
        }
        s1 = 0u;
    };

    acc_t prev0, prev1;
#pragma unroll
    for (int r = 0; r < 16; ++r) prev0[r] = prev1[r] = 0;

    // fragment ring registers: slot s of a step consumes r0/r1[s % PF]
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    uint4 r0[PF], r1[PF];
    {
        const uint4* b0 = reinterpret_cast<const uint4*>(smem) + lane;
        const uint4* b1 = reinterpret_cast<const uint4*>(smem + TILE_BYTES) + lane;
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            r0[s] = b0[s * 64];
            r1[s] = b1[s * 64];
        }
    }
    unsigned ring = 0u;  // ring slot of tile `it` (even, 0 .. NBUF - 2)

    auto do_step = [&](int it, auto Hc) {
        constexpr int H = decltype(Hc)::value;  // which half of the 4-tile chunk
        // every wave's pieces of tiles it+2, it+3 (issued during the previous step) have landed
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned ring2 = ring + 2u >= (unsigned)NBUF ? ring + 2u - NBUF : ring + 2u;   // tiles it+2, it+3
        const unsigned ring4 = ring2 + 2u >= (unsigned)NBUF ? ring2 + 2u - NBUF : ring2 + 2u;  // tiles it+4, it+5
        const uint4* cur0 = reinterpret_cast<const uint4*>(smem + ring * TILE_BYTES) + lane;
        const uint4* cur1 = cur0 + TILE_U4;
        const uint4* nxt0 = reinterpret_cast<const uint4*>(smem + ring2 * TILE_BYTES) + lane;
        const uint4* nxt1 = nxt0 + TILE_U4;
        acc_t acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = COARSE_OFFSET;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&r0[s % PF]), qf[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&r1[s % PF]), qf[s], acc1, 0, 0, 0);
            if (s + PF < KSTEPS) {
                r0[s % PF] = cur0[(s + PF) * 64];
                r1[s % PF] = cur1[(s + PF) * 64];
            } else {  // first fragments of the next step (stale data after the last step: unused)
                r0[s % PF] = nxt0[(s + PF - KSTEPS) * 64];
                r1[s % PF] = nxt1[(s + PF - KSTEPS) * 64];
            }
            // deferred fold of the previous step's two tiles, spread over the slots
#pragma unroll
            for (int e = s * 32 / KSTEPS; e < (s + 1) * 32 / KSTEPS; ++e)
                fold_one(e < 16 ? prev0[e & 15] : prev1[e & 15], (2 * (1 - H) + (e >> 4)) * 16 + (e & 15));
            if (s >= 1 && s <= PIECES) {  // one 1 KiB piece per slot instead of a burst in slot 1
                __builtin_amdgcn_sched_barrier(0);
                if (it + 4 < ntiles) {
                    const int p = s - 1;
                    glds16(gnext + p * NWAVES * 64,
                           __builtin_amdgcn_readfirstlane(ldst0 + ring4 * TILE_BYTES + (unsigned)(p * NWAVES) * 1024u));
                }
                if (s == PIECES) gnext += 2 * TILE_U4;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SPARSE) {
            if (it >= 2) step_tail(it - 2, prev0, prev1);
            else s1 = 0u;
        } else {
            if constexpr (H == 0) fold_tail(it - 1);
        }
        prev0 = acc0;
        prev1 = acc1;
        ring = ring2;
    };

    for (int it = 0; it < ntiles; it += 4) {
        do_step(it, std::integral_constant<int, 0>{});
        do_step(it + 2, std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) fold_one(e < 16 ? prev0[e & 15] : prev1[e & 15], (2 + (e >> 4)) * 16 + (e & 15));
    if constexpr (SPARSE) step_tail(ntiles - 2, prev0, prev1);
    else fold_tail(ntiles - 1);
    if (lane < 32 && qt < a.nq_tiles) atomicMax(a.qmax + (size_t)qt * 32 + lane, runmax);
    if constexpr (SPARSE) {  // flush the workgroup's records to the per-query lists (no DMA in flight any more)
        __syncthreads();
        const unsigned cnt = min(*lrec_count, (unsigned)LREC_CAP);
        for (unsigned i = threadIdx.x; i < cnt; i += 512) {
            const uint2 r = lrec[i];
            const size_t qi = (size_t)qb * QBLOCK + (r.x >> 24);
            const unsigned gs = atomicAdd(a.rec_cnt + qi, 1u);
            if (gs < (unsigned)a.rcap) a.rec[qi * (size_t)a.rcap + gs] = make_uint2(r.x & 0xFFFFFFu, r.y);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// int8 coarse pass with T map tiles per step (the schedule of match_coarse_pipe_kernel<., false, true>, generalised).
// One barrier and one staging round per T * KSTEPS MFMAs: at d = 768 (KSTEPS = 24, T = 2: 48 MFMAs per step) the int8
// kernel measured 0.53 of the int8 peak, at d = 384 (KSTEPS = 12, T = 2: 24 per step) 0.49 -- the per-step cost is fixed,
// so d = 384 / 256 take T = 4 here (one whole 128-row chunk per step, 48 / 32 MFMAs).  Ring = 3 steps of T tiles
// (in use | landed | in flight), fragment look-ahead PF = 2 k-steps for T = 4 (4 tiles x 2 x 4 registers), 4 for T = 2.
// ---------------------------------------------------------------------------------------------
// TOP2: the packed per-chunk top-2 records of the fp16 pass (best row index included, 3 VALU ops per element) instead of the
// best value alone: candidate chunks with one row inside the bounds need no rescan -- the choice for duplicate-rich maps.
template <int KSTEPS, int T, bool TOP2 = false>
__global__ __launch_bounds__(512, 2) void match_coarse_i8_kernel(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVES = 8;
    constexpr int TILE_U4 = KSTEPS * 64;
    constexpr int TILE_BYTES = TILE_U4 * 16;
    constexpr int PIECES = T * KSTEPS / NWAVES;  // 1 KiB pieces per wave per step
    constexpr int NBUF = 3 * T;
    constexpr int PF = T == 4 ? 2 : 4;
    static_assert(T == 2 || T == 4, "a 128-row chunk is 4 tiles");
    static_assert((T * KSTEPS) % NWAVES == 0 && PIECES < KSTEPS, "a step's tiles must split evenly over the waves");
    static_assert(KSTEPS % PF == 0 && KSTEPS >= 2 * PF, "fragment ring must align across steps");
    static_assert(NBUF * TILE_BYTES <= 160 * 1024, "ring exceeds the LDS");

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const CoarseUnit cu = coarse_unit(a);
    const int qb = cu.qb, c0 = cu.c0, ntiles = cu.ntiles;  // ntiles: a multiple of 4 (whole chunks)
    if (ntiles == 0) return;
    const int qt = qb * NWAVES + wave;  // this wave's 32-query tile

    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS unsigned char*)smem;
    const uint4* gsrc = a.Bh + (size_t)c0 * 4 * TILE_U4 + wave * 64 + lane;  // piece p of a step: + p * NWAVES * 64
    const unsigned ldst0 = lds_base + (unsigned)wave * 1024u;
    auto stage_step = [&](const uint4* src, unsigned ring_byte) {  // T consecutive tiles: contiguous in memory and in the ring
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            glds16(src + p * NWAVES * 64, __builtin_amdgcn_readfirstlane(ldst0 + ring_byte + (unsigned)(p * NWAVES) * 1024u));
    };

    intx4 qf[KSTEPS];
    {
        const uint4* qsrc = a.Qh + (size_t)(qt < a.nq_tiles ? qt : 0) * TILE_U4 + lane;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            uint4 v = qsrc[s * 64];
            qf[s] = *reinterpret_cast<intx4*>(&v);
        }
    }
    stage_step(gsrc, 0u);
    if (ntiles > T) stage_step(gsrc + (size_t)T * TILE_U4, (unsigned)(T * TILE_BYTES));
    const uint4* gnext = gsrc + (size_t)2 * T * TILE_U4;  // next step to stage

    // per-lane constants of the query's bound and the running lower bound of its exact maximum (match_select_kernel)
    float i8_sq, i8_A, i8_mult, i8_low = -__builtin_inff();
    {
        const size_t qi = (size_t)(qt < a.nq_tiles ? qt : 0) * 32 + (lane & 31);
        const float eq = a.ib.qerr[qi];
        i8_sq = a.ib.qstep[qi >> 7];
        i8_A = eq * 1.0001220703125f + 1.0e-6f;
        i8_mult = 1.0001220703125f + eq;
    }
    unsigned s1 = 0u, s2 = 0u, unused_max = 0u;
    auto emit_chunk = [&](int chunk) __attribute__((always_inline)) {  // chunk < 0: nothing folded yet
        const unsigned best = TOP2 ? coarse_emit_chunk(a, s1, s2, unused_max, qt, chunk) : coarse_emit_chunk_best(a, s1, qt, chunk);
        if (chunk >= 0 && chunk < a.first_pad_chunk) {  // wave-uniform
            const float sb = a.ib.bstep[chunk], be = a.ib.berr[chunk];
            i8_low = fmaxf(i8_low, __builtin_fmaf(i8_sq * sb, (float)((int)best - I8_OFFSET), -(i8_A + i8_mult * be)));
        }
    };

    intx16 prev[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[t][r] = 0;

    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    uint4 fr[T][PF];  // fragment ring registers: slot s of a step consumes fr[t][s % PF]
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint4* b0 = reinterpret_cast<const uint4*>(smem + t * TILE_BYTES) + lane;
#pragma unroll
        for (int s = 0; s < PF; ++s) fr[t][s] = b0[s * 64];
    }
    unsigned ring = 0u;  // ring slot of the step's first tile (0, T, 2T)

    for (int it = 0; it < ntiles; it += T) {
        // every wave's pieces of the next step (issued during the previous step) have landed
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned ring1 = ring + T >= (unsigned)NBUF ? ring + T - NBUF : ring + T;      // next step
        const unsigned ring2 = ring1 + T >= (unsigned)NBUF ? ring1 + T - NBUF : ring1 + T;   // the step after: being refilled
        const uint4* cur = reinterpret_cast<const uint4*>(smem + ring * TILE_BYTES) + lane;
        const uint4* nxt = reinterpret_cast<const uint4*>(smem + ring1 * TILE_BYTES) + lane;
        intx16 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = I8_OFFSET;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
            for (int t = 0; t < T; ++t)
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<intx4*>(&fr[t][s % PF]), qf[s], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (s + PF < KSTEPS) fr[t][s % PF] = cur[t * TILE_U4 + (s + PF) * 64];
                else fr[t][s % PF] = nxt[t * TILE_U4 + (s + PF - KSTEPS) * 64];  // first fragments of the next step (stale after the last)
            }
            // deferred fold of the previous step's tiles, spread over the slots
#pragma unroll
            for (int e = s * 16 * T / KSTEPS; e < (s + 1) * 16 * T / KSTEPS; ++e) {
                if constexpr (TOP2) coarse_fold(s1, s2, prev[e >> 4][e & 15], (((it - T) & 3) + (e >> 4)) * 16 + (e & 15));
                else s1 = max(s1, (unsigned)prev[e >> 4][e & 15]);
            }
            if (s >= 1 && s <= PIECES) {  // one 1 KiB piece per slot instead of a burst
                __builtin_amdgcn_sched_barrier(0);
                if (it + 2 * T < ntiles) {
                    const int p = s - 1;
                    glds16(gnext + p * NWAVES * 64,
                           __builtin_amdgcn_readfirstlane(ldst0 + ring2 * TILE_BYTES + (unsigned)(p * NWAVES) * 1024u));
                }
                if (s == PIECES) gnext += T * TILE_U4;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the tiles folded in this step were it - T .. it - 1: a chunk is complete when `it` is a multiple of 4
        if ((it & 3) == 0) emit_chunk(it >= 4 ? c0 + (it >> 2) - 1 : -1);
#pragma unroll
        for (int t = 0; t < T; ++t) prev[t] = acc[t];
        ring = ring1;
    }
#pragma unroll
    for (int e = 0; e < 16 * T; ++e) {
        if constexpr (TOP2) coarse_fold(s1, s2, prev[e >> 4][e & 15], ((4 - T) + (e >> 4)) * 16 + (e & 15));
        else s1 = max(s1, (unsigned)prev[e >> 4][e & 15]);
    }
    emit_chunk(c0 + (ntiles >> 2) - 1);
    if (lane < 32 && qt < a.nq_tiles) atomicMax(a.qmax + (size_t)qt * 32 + lane, float_key(i8_low));
}

// ---------------------------------------------------------------------------------------------
// int8 coarse pass, 64 resident queries per wave (d = 256 / 384: 2 x 48 query registers fit beside everything else).
// What the fp16 kernel could not afford (2 x 96 query registers) and its ablations named as the remaining cost: every map
// fragment read from LDS feeds TWO MFMAs and a workgroup covers 512 queries, so LDS operand reads and L2 -> LDS staging per
// MFMA both halve.  The four tiles of a step are processed one after the other (a tile = KSTEPS slots of two MFMAs, one per
// query set, sharing the fragment); the accumulators of a finished tile are folded in the slots of the next one, and two
// accumulator pairs alternate, so nothing is copied.  One barrier per 8 * KSTEPS MFMAs.
// ---------------------------------------------------------------------------------------------
template <int KSTEPS, bool TOP2 = false>
__global__ __launch_bounds__(512, 2) void match_coarse_i8q2_kernel(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVES = 8, T = 4;
    constexpr int TILE_U4 = KSTEPS * 64;
    constexpr int TILE_BYTES = TILE_U4 * 16;
    constexpr int PIECES = T * KSTEPS / NWAVES;  // 1 KiB pieces per wave per step: 6 (d = 384) or 4 (d = 256)
    constexpr int NBUF = 3 * T;
    constexpr int PF = 4;
    static_assert(KSTEPS % PF == 0 && KSTEPS >= 2 * PF && PIECES <= 2 * T, "shape");
    static_assert(NBUF * TILE_BYTES <= 160 * 1024, "ring exceeds the LDS");

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const CoarseUnit cu = coarse_unit(a);
    const int qb = cu.qb, c0 = cu.c0, ntiles = cu.ntiles;  // ntiles: a multiple of 4 (whole chunks)
    if (ntiles == 0) return;
    const int qt0 = (qb * NWAVES + wave) * 2;  // this wave's two 32-query tiles

    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS unsigned char*)smem;
    const uint4* gsrc = a.Bh + (size_t)c0 * 4 * TILE_U4 + wave * 64 + lane;
    const unsigned ldst0 = lds_base + (unsigned)wave * 1024u;
    auto stage_step = [&](const uint4* src, unsigned ring_byte) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            glds16(src + p * NWAVES * 64, __builtin_amdgcn_readfirstlane(ldst0 + ring_byte + (unsigned)(p * NWAVES) * 1024u));
    };

    intx4 qf[2][KSTEPS];
    float i8_sq[2], i8_A[2], i8_mult[2], i8_low[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qt = qt0 + j < a.nq_tiles ? qt0 + j : 0;
        const uint4* qsrc = a.Qh + (size_t)qt * TILE_U4 + lane;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            uint4 v = qsrc[s * 64];
            qf[j][s] = *reinterpret_cast<intx4*>(&v);
        }
        const size_t qi = (size_t)qt * 32 + (lane & 31);
        const float eq = a.ib.qerr[qi];
        i8_sq[j] = a.ib.qstep[qi >> 7];
        i8_A[j] = eq * 1.0001220703125f + 1.0e-6f;
        i8_mult[j] = 1.0001220703125f + eq;
        i8_low[j] = -__builtin_inff();
    }
    stage_step(gsrc, 0u);
    if (ntiles > T) stage_step(gsrc + (size_t)T * TILE_U4, (unsigned)(T * TILE_BYTES));
    const uint4* gnext = gsrc + (size_t)2 * T * TILE_U4;

    unsigned s1[2] = {0u, 0u}, s2[2] = {0u, 0u}, unused_max = 0u;
    auto emit_chunk = [&](int chunk) __attribute__((always_inline)) {  // chunk < 0: nothing folded yet
        float sb = 0.f, be = 0.f;
        const bool counted = chunk >= 0 && chunk < a.first_pad_chunk;  // wave-uniform
        if (counted) {
            sb = a.ib.bstep[chunk];
            be = a.ib.berr[chunk];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned best = TOP2 ? coarse_emit_chunk(a, s1[j], s2[j], unused_max, qt0 + j, chunk)
                                       : coarse_emit_chunk_best(a, s1[j], qt0 + j, chunk);
            if (counted)
                i8_low[j] = fmaxf(i8_low[j], __builtin_fmaf(i8_sq[j] * sb, (float)((int)best - I8_OFFSET), -(i8_A[j] + i8_mult[j] * be)));
        }
    };

    intx16 accA[2], accB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[j][r] = accB[j][r] = 0;

    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    uint4 fr[PF];  // fragment ring registers of the tile in progress: slot s consumes fr[s % PF]
    {
        const uint4* b0 = reinterpret_cast<const uint4*>(smem) + lane;
#pragma unroll
        for (int s = 0; s < PF; ++s) fr[s] = b0[s * 64];
    }
    unsigned ring = 0u;

    for (int it = 0; it < ntiles; it += T) {
        wait_vmcnt<0>();  // every wave's pieces of the next step (issued during the previous step) have landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned ring1 = ring + T >= (unsigned)NBUF ? ring + T - NBUF : ring + T;
        const unsigned ring2 = ring1 + T >= (unsigned)NBUF ? ring1 + T - NBUF : ring1 + T;
        const uint4* cur = reinterpret_cast<const uint4*>(smem + ring * TILE_BYTES) + lane;
        const uint4* nxt = reinterpret_cast<const uint4*>(smem + ring1 * TILE_BYTES) + lane;
        const bool more = it + 2 * T < ntiles;
        // tile J of the step into `acc`, folding `done` (the tile before it)
        auto tile = [&](auto Jc, intx16 (&acc)[2], const intx16 (&done)[2]) __attribute__((always_inline)) {
            constexpr int J = decltype(Jc)::value;
            const uint4* tb = cur + J * TILE_U4;
            const uint4* tn = (J + 1 < T) ? cur + (J + 1) * TILE_U4 : nxt;  // the tile after it (stale after the last step)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = I8_OFFSET;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<intx4*>(&fr[s % PF]), qf[0][s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<intx4*>(&fr[s % PF]), qf[1][s], acc[1], 0, 0, 0);
                fr[s % PF] = (s + PF < KSTEPS) ? tb[(s + PF) * 64] : tn[(s + PF - KSTEPS) * 64];
#pragma unroll
                for (int e = s * 32 / KSTEPS; e < (s + 1) * 32 / KSTEPS; ++e) {  // `done` is tile (J + 3) & 3 of its chunk
                    if constexpr (TOP2) coarse_fold(s1[e >> 4], s2[e >> 4], done[e >> 4][e & 15], ((J + 3) & 3) * 16 + (e & 15));
                    else s1[e >> 4] = max(s1[e >> 4], (unsigned)done[e >> 4][e & 15]);
                }
                if (s >= 1 && s <= 2 && J * 2 + s - 1 < PIECES) {  // two 1 KiB pieces per tile
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        const int p = J * 2 + s - 1;
                        glds16(gnext + p * NWAVES * 64,
                               __builtin_amdgcn_readfirstlane(ldst0 + ring2 * TILE_BYTES + (unsigned)(p * NWAVES) * 1024u));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        tile(std::integral_constant<int, 0>{}, accA, accB);
        emit_chunk(it >= 4 ? c0 + (it >> 2) - 1 : -1);  // tile 0's slots folded the last tile of the previous chunk
        tile(std::integral_constant<int, 1>{}, accB, accA);
        tile(std::integral_constant<int, 2>{}, accA, accB);
        tile(std::integral_constant<int, 3>{}, accB, accA);
        gnext += T * TILE_U4;
        ring = ring1;
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        if constexpr (TOP2) coarse_fold(s1[e >> 4], s2[e >> 4], accB[e >> 4][e & 15], 3 * 16 + (e & 15));
        else s1[e >> 4] = max(s1[e >> 4], (unsigned)accB[e >> 4][e & 15]);
    }
    emit_chunk(c0 + (ntiles >> 2) - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (lane < 32 && qt0 + j < a.nq_tiles) atomicMax(a.qmax + (size_t)(qt0 + j) * 32 + lane, float_key(i8_low[j]));
}

// ---------------------------------------------------------------------------------------------
// coarse pass, 4 waves per workgroup with QSETS x 32 resident queries each ("register-heavy")
//
// One wave per SIMD and the unified 512-register budget: a wave keeps QSETS x 32 queries resident
// (QSETS * d/4 registers) and every map fragment read from LDS feeds QSETS MFMAs -- LDS operand
// traffic per flop / QSETS, and with QSETS = 3 a workgroup covers 384 queries, so the L2 -> LDS staging
// per flop drops to 2/3 as well (the two costs the ablations of the 8-wave kernel measured).
// Measured at d = 384 (C2): QSETS = 2 -> 3.26 ms, QSETS = 3 -> register spills, vs 2.6 ms for the 8-wave
// kernel: one wave per SIMD cannot slot the fold / LDS / scalar stream between its own MFMAs as well as
// two waves hide each other, so this shape is instantiated only where it is the only one that fits:
// wide descriptors (d = 640, 768: config C5, QSETS = 1, 192 query registers), 1.04-1.06 PFLOP/s.  A step is ONE 32-row map tile (d/16 KiB in LDS): tile i in use, i+1 landed (its first
// fragments are prefetched across the barrier), i+2 .. i+NBUF-2 in flight, the slot of i-1 is being
// refilled.  QSETS = 1 alternates the k-steps between two accumulator chains (added before the fold);
// QSETS >= 2 has one chain per query set.  Same packed top-2 records, select / rescore as every other
// variant.  Error bound of the coarse score at d = 768: 48 instead of 24 accumulation steps add
// < 5e-5, E < 1.15e-3, window 2.5e-3 >= 2E still holds.
// ---------------------------------------------------------------------------------------------
template <int KSTEPS, int QSETS, int NBUF, bool BIAS>
__global__ __launch_bounds__(256, 1) void match_coarse_r_kernel(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVES = 4;
    constexpr int TILE_U4 = KSTEPS * 64;
    constexpr int TILE_BYTES = TILE_U4 * 16;
    constexpr int PASSES = KSTEPS / NWAVES;
    constexpr int PF = 4;
    constexpr int SPLIT = (QSETS == 1) ? 2 : 1;  // accumulator chains per query set
    constexpr int FOLD = 16 * QSETS;             // accumulator elements folded per step
    static_assert(KSTEPS % 8 == 0 && KSTEPS % PF == 0, "d must be a multiple of 128");
    static_assert(NBUF >= 3 && NBUF * TILE_BYTES <= 160 * 1024, "ring must fit the LDS");
    static_assert(QSETS * KSTEPS * 4 <= 320, "resident queries must leave registers for the accumulators");

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const CoarseUnit cu = coarse_unit(a);
    const int qb = cu.qb, c0 = cu.c0, ntiles = cu.ntiles;
    const int qt0 = (qb * NWAVES + wave) * QSETS;  // first 32-query tile of this wave

    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS unsigned char*)smem;
    const uint4* gsrc = a.Bh + (size_t)c0 * 4 * TILE_U4 + wave * 64 + lane;
    const unsigned ldst0 = lds_base + (unsigned)wave * 1024u;
    auto stage = [&](const uint4* src, unsigned ring_byte) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
            glds16(src + p * NWAVES * 64, __builtin_amdgcn_readfirstlane(ldst0 + ring_byte + (unsigned)(p * NWAVES) * 1024u));
    };

    half8 qf[QSETS][KSTEPS];
#pragma unroll
    for (int j = 0; j < QSETS; ++j) {
        const int qt = qt0 + j;
        const uint4* qsrc = a.Qh + (size_t)(qt < a.nq_tiles ? qt : 0) * TILE_U4 + lane;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            uint4 v = qsrc[s * 64];
            qf[j][s] = *reinterpret_cast<half8*>(&v);
        }
    }
    // tiles 0 .. NBUF-2 go out before the loop (a slice has >= 4 tiles; clamp for tiny ones)
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (i < ntiles) stage(gsrc + (size_t)i * TILE_U4, (unsigned)(i * TILE_BYTES));
    const uint4* gnext = gsrc + (size_t)(NBUF - 1) * TILE_U4;

    unsigned s1[QSETS], s2[QSETS], runmax[QSETS];
#pragma unroll
    for (int j = 0; j < QSETS; ++j) s1[j] = s2[j] = runmax[j] = 0u;
    auto fold_tail = [&](int it, auto Jc) {
        constexpr int J = decltype(Jc)::value;
        coarse_emit_chunk(a, s1[J], s2[J], runmax[J], qt0 + J, it >= 0 ? c0 + (it >> 2) : -1);
    };

    floatx16 prev[QSETS];
#pragma unroll
    for (int j = 0; j < QSETS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[j][r] = 0.f;

    // all of tile 0 and this wave's pieces of tile 1 .. : wait for everything once
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    uint4 rf[PF];
    {
        const uint4* b0 = reinterpret_cast<const uint4*>(smem) + lane;
#pragma unroll
        for (int s = 0; s < PF; ++s) rf[s] = b0[s * 64];
    }
    unsigned ring = 0u;  // ring slot of tile `it`

    auto do_step = [&](int it, auto Pc) {
        constexpr int P = decltype(Pc)::value;  // tile index inside the 4-tile chunk
        // tile it+1 must have landed; tiles it+2 .. it+NBUF-2 may stay in flight
        if (NBUF > 3 && it + 2 < ntiles) wait_vmcnt<(NBUF - 3) * PASSES>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned ring1 = ring + 1u >= (unsigned)NBUF ? 0u : ring + 1u;
        const unsigned ringf = ring == 0u ? (unsigned)NBUF - 1u : ring - 1u;  // slot of tile it-1 = tile it+NBUF-1
        const uint4* cur = reinterpret_cast<const uint4*>(smem + ring * TILE_BYTES) + lane;
        const uint4* nxt = reinterpret_cast<const uint4*>(smem + ring1 * TILE_BYTES) + lane;
        floatx16 acc[QSETS][SPLIT];
#pragma unroll
        for (int j = 0; j < QSETS; ++j)
#pragma unroll
            for (int c = 0; c < SPLIT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][c][r] = (c == 0) ? COARSE_OFFSET : 0.f;
        if constexpr (BIAS) {
            // accumulator register r of a lane holds map row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the tile
            const float* bt = a.row_bias + ((size_t)c0 * 4 + (size_t)it) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(bt + 8 * g);
#pragma unroll
                for (int j = 0; j < QSETS; ++j) {
                    acc[j][0][4 * g + 0] += bv.x;
                    acc[j][0][4 * g + 1] += bv.y;
                    acc[j][0][4 * g + 2] += bv.z;
                    acc[j][0][4 * g + 3] += bv.w;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
            for (int j = 0; j < QSETS; ++j)
                acc[j][s % SPLIT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<half8*>(&rf[s % PF]), qf[j][s],
                                                                           acc[j][s % SPLIT], 0, 0, 0);
            rf[s % PF] = (s + PF < KSTEPS) ? cur[(s + PF) * 64] : nxt[(s + PF - KSTEPS) * 64];
            // deferred fold of the previous tile (tile (P + 3) % 4 of its chunk), spread over the slots
#pragma unroll
            for (int e = s * FOLD / KSTEPS; e < (s + 1) * FOLD / KSTEPS; ++e) {
                const int j = e >> 4, r = e & 15;
                coarse_fold(s1[j], s2[j], prev[j][r], ((P + 3) & 3) * 16 + r);
            }
            if (s >= 1 && s <= PASSES) {  // one 1 KiB piece of tile it+NBUF-1 per slot (no burst after the barrier)
                __builtin_amdgcn_sched_barrier(0);
                if (it + NBUF - 1 < ntiles)  // uniform
                    glds16(gnext + (s - 1) * NWAVES * 64,
                           __builtin_amdgcn_readfirstlane(ldst0 + ringf * TILE_BYTES + (unsigned)((s - 1) * NWAVES) * 1024u));
                if (s == PASSES) gnext += TILE_U4;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (P == 0) {
            fold_tail(it - 1, std::integral_constant<int, 0>{});
            if constexpr (QSETS > 1) fold_tail(it - 1, std::integral_constant<int, 1>{});
            if constexpr (QSETS > 2) fold_tail(it - 1, std::integral_constant<int, 2>{});
        }
#pragma unroll
        for (int j = 0; j < QSETS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) prev[j][r] = (SPLIT == 2) ? acc[j][0][r] + acc[j][SPLIT - 1][r] : acc[j][0][r];
        ring = ring1;
    };

    for (int it = 0; it < ntiles; it += 4) {
        do_step(it, std::integral_constant<int, 0>{});
        do_step(it + 1, std::integral_constant<int, 1>{});
        do_step(it + 2, std::integral_constant<int, 2>{});
        do_step(it + 3, std::integral_constant<int, 3>{});
    }
#pragma unroll
    for (int j = 0; j < QSETS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) coarse_fold(s1[j], s2[j], prev[j][r], 3 * 16 + r);
    fold_tail(ntiles - 1, std::integral_constant<int, 0>{});
    if constexpr (QSETS > 1) fold_tail(ntiles - 1, std::integral_constant<int, 1>{});
    if constexpr (QSETS > 2) fold_tail(ntiles - 1, std::integral_constant<int, 2>{});
#pragma unroll
    for (int j = 0; j < QSETS; ++j)
        if (lane < 32 && qt0 + j < a.nq_tiles) atomicMax(a.qmax + (size_t)(qt0 + j) * 32 + lane, runmax[j]);
}

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
                    rec[u] = (c < nchunks) ? partials[(size_t)c * npad + q] : make_uint2(0u, 0u);
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
struct RefineWave {
    const float* b;
    const float* invb;
    int d, nt, g, l, lane;
    float w2;
    float4 qv[12];
    unsigned* lrow;
    float* lsc;
    int kept;        // wave-uniform
    float runmax;    // wave-uniform
    bool overflow;

    __device__ __forceinline__ void init(const float* q, float iq, int64_t qi, const float* b_, const float* invb_, int d_, float w2_,
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
        for (int t = 0; t < 12; ++t) {
            if (t < nt) {
                float4 v = *reinterpret_cast<const float4*>(q + qi * (int64_t)d + 4 * (l + 16 * t));
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
            const float* br = b + row * (int64_t)d + 4 * l;
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                if (t < nt) {
                    const float4 bv = *reinterpret_cast<const float4*>(br + 64 * t);
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
                                                           int* __restrict__ fb_list, int* __restrict__ todo) {
    __shared__ uint4 l_q8[4][48];  // the query's int8 row, unit by unit (d <= 768)
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
    if (qi >= n) return;
    const int cnt = cand_cnt[qi];
    if (cnt <= 0) return;  // zero query / below the gate (-2) / overflow (-1)
    unsigned* mycand = cand + (size_t)qi * cap;
    unsigned* myhits = hits + (size_t)qi * hcap;
    const int units8 = d >> 4;  // 16-byte units per int8 row
    if (lane < units8) l_q8[wave][lane] = q8[(size_t)(qi >> 5) * (units8 * 32) + (size_t)lane * 32 + (qi & 31)];
    const float eq = ib.qerr[qi];
    const float sq = ib.qstep[qi >> 7], A = eq * 1.0001220703125f + 1.0e-6f, mult = 1.0001220703125f + eq;
    const float qlow = key_float(qmax[qi]);
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
                const bool hit = base + rr < m && sc * (float)acc + bound >= qlow;
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
        // crowded: match_refine_kernel's work list (up to seven rows go straight to the fp64 decision: a handful of fp64
        // dot products costs less than the latency of one refinement wave)
        if (nhit >= REFINE_MIN_I8) todo[atomicAdd(fb_count + 6, 1)] = (int)qi;
    }
}

// dense records: the candidate entries of match_select_kernel (int8 pass: as rewritten by match_rescan_kernel)
__global__ __launch_bounds__(256) void match_refine_kernel(const float* __restrict__ q, const float* __restrict__ invq,
                                                           const float* __restrict__ b, const float* __restrict__ invb,
                                                           int64_t n, int64_t m, int d, float w2, int* __restrict__ cand_cnt,
                                                           unsigned* __restrict__ cand, int cap, int* __restrict__ fb_count,
                                                           int* __restrict__ fb_list, int stats, const int* __restrict__ todo,
                                                           const int* __restrict__ todo_count) {
    __shared__ unsigned l_row[4][REFINE_KEEP];
    __shared__ float l_sc[4][REFINE_KEEP];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    // todo != NULL (int8 pass): the queries match_rescan_kernel found crowded, a short list walked by a small grid;
    // otherwise one wave per query
    const int64_t slot0 = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nslots = todo ? (int64_t)*todo_count : n;
  for (int64_t slot = slot0; slot < nslots; slot += (int64_t)gridDim.x * 4) {
    const int64_t qi = todo ? (int64_t)todo[slot] : slot;
    if (qi >= n) return;
    const int cnt = cand_cnt[qi];
    if (cnt <= 0) continue;  // zero query / nothing / overflow (-1: the all-pairs kernel decides)
    unsigned* mycand = cand + (size_t)qi * cap;
    // wave-uniform: is this list crowded?
    bool flagged = false;
    for (int e = lane; e < cnt; e += 64) flagged |= (mycand[e] & 128u) != 0u;
    if (cnt < REFINE_MIN && !__any(flagged)) continue;
    RefineWave R;
    R.init(q, invq[qi], qi, b, invb, d, w2, l_row[wave], l_sc[wave]);
    // single-row entries: 4 per pass
    for (int e0 = 0; e0 < cnt; e0 += 4) {
        long long row = -1;
        if (e0 + R.g < cnt) {
            const unsigned ce = mycand[e0 + R.g];
            if (!(ce & 128u)) {
                row = (long long)(ce >> 8) * CHUNK_ROWS + (ce & 127u);
                if (row >= m) row = -1;
            }
        }
        if (__any(row >= 0)) R.consider(row, R.score4(row));
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
constexpr int FILTER_LDS_ROWS = 1024;  // >= rcap
__global__ __launch_bounds__(256) void match_filter_refine_kernel(const float* __restrict__ q, const float* __restrict__ invq,
                                                                  const float* __restrict__ b, const float* __restrict__ invb,
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
    RefineWave R;
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
__device__ __forceinline__ double dot_norm_f64(const float* __restrict__ qrow_n, const float* __restrict__ brow, float invb, int d) {
    double acc = 0.0;
    for (int k = 0; k < d; k += 4) {
        const float4 bv = *reinterpret_cast<const float4*>(brow + k);
        const float b0 = bv.x * invb, b1 = bv.y * invb, b2 = bv.z * invb, b3 = bv.w * invb;
        acc = acc + (double)qrow_n[k + 0] * (double)b0;
        acc = acc + (double)qrow_n[k + 1] * (double)b1;
        acc = acc + (double)qrow_n[k + 2] * (double)b2;
        acc = acc + (double)qrow_n[k + 3] * (double)b3;
    }
    return acc;
}

__device__ __forceinline__ void wave_argmax(double& s, long long& j) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double so = __shfl_xor(s, off);
        const long long jo = __shfl_xor(j, off);
        if (jo >= 0 && (j < 0 || so > s || (so == s && jo < j))) {
            s = so;
            j = jo;
        }
    }
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
__global__ __launch_bounds__(256) void match_rescore_kernel(const float* __restrict__ q, const float* __restrict__ invq,
                                                            const float* __restrict__ b, const float* __restrict__ invb,
                                                            int64_t n, int64_t m, int d, const int* __restrict__ cand_cnt,
                                                            const unsigned* __restrict__ cand, int cap,
                                                            int64_t* __restrict__ idx_out, float* __restrict__ sim_out) {
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
                        qv[it] = *reinterpret_cast<const float4*>(q + (q0 + s_ql[sl]) * (int64_t)d + k0 + 4 * l4);
                        bv[it] = *reinterpret_cast<const float4*>(b + jj * (int64_t)d + k0 + 4 * l4);
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
                float4 v = *reinterpret_cast<const float4*>(q + qq * (int64_t)d + k);
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
                        const double sc = dot_norm_f64(qn, b + j * (int64_t)d, invb[j], d);
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
        idx_out[qi] = bj;
        sim_out[qi] = (float)best;
    } else if (cand_cnt[qi] == -2) {  // below the caller's gate (match_select_kernel)
        idx_out[qi] = -1;
        sim_out[qi] = -2.0f;
    }
}

// all-pairs exact decision for the queries in `list` (or all queries if list == NULL)
__global__ __launch_bounds__(256) void match_exact_kernel(const float* __restrict__ q, const float* __restrict__ invq,
                                                          const float* __restrict__ b, const float* __restrict__ invb,
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
        for (int k = threadIdx.x; k < d; k += 256) qn[k] = q[qi * (int64_t)d + k] * iq;
        __syncthreads();
        double best = 0.0;
        long long bj = -1;
        for (long long j = threadIdx.x; j < m; j += 256) {
            const double s = dot_norm_f64(qn, b + j * (int64_t)d, invb ? invb[j] : 1.0f, d);
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
__global__ __launch_bounds__(1024) void threshold_compact_kernel(const float* __restrict__ sim, const int64_t* __restrict__ idx,
                                                                 int64_t n, double thr, int64_t* __restrict__ keep,
                                                                 int64_t* __restrict__ count, int32_t* __restrict__ corres,
                                                                 const double* __restrict__ qxyz, const double* __restrict__ bxyz,
                                                                 double* __restrict__ src_out, double* __restrict__ tgt_out) {
    __shared__ int wsum[16];
    __shared__ int64_t base_s;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int64_t s = 0; s < n; s += 1024) {
        const int64_t i = s + threadIdx.x;
        const bool valid = (i < n) && !((double)sim[i] < thr);
        const unsigned long long bal = __ballot(valid);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) woff += wsum[w];
            tot += wsum[w];
        }
        const int64_t base = base_s;
        if (valid) {
            const int64_t k = base + woff + before;
            keep[k] = i;
            const int64_t j = idx ? idx[i] : 0;
            if (corres) {
                corres[2 * k + 0] = (int32_t)i;
                corres[2 * k + 1] = (int32_t)j;
            }
            if (src_out) {
                src_out[3 * k + 0] = qxyz[3 * i + 0];
                src_out[3 * k + 1] = qxyz[3 * i + 1];
                src_out[3 * k + 2] = qxyz[3 * i + 2];
            }
            if (tgt_out) {
                tgt_out[3 * k + 0] = bxyz[3 * j + 0];
                tgt_out[3 * k + 1] = bxyz[3 * j + 1];
                tgt_out[3 * k + 2] = bxyz[3 * j + 2];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s = base + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base_s;
}

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

// =============================================================================================
// host side: C ABI (include/vfmreg.h)
// =============================================================================================
namespace {

struct Prepared {
    float* inv;
    uint4* tiles;
    // int8 image (i8_capable(d); prep_chunk_kernel)
    float* err;       // E per row
    float* gstep;     // quantisation step per group of 128 rows
    float* gerr;      // maximum E per group
    uint4* tiles8;    // int8 fragment tiles
    size_t bytes;
};

// widths the int8 coarse pass exists for (match_coarse_pipe_kernel<d/32, false, true>; d = 128 has too few k-steps for the
// fragment ring)
inline bool i8_capable(int d) { return d == 256 || d == 384 || d == 512 || d == 640 || d == 768; }

inline Prepared carve_prepared(void* p, int64_t rows, int d) {
    VfmCarver c(p);
    Prepared r;
    const int64_t rp = rows_padded(rows);
    r.inv = c.take<float>((size_t)rp);
    r.tiles = c.take<uint4>((size_t)rp / TILE_ROWS * (size_t)(d / 16) * 64);
    r.err = nullptr;
    r.gstep = r.gerr = nullptr;
    r.tiles8 = nullptr;
    if (i8_capable(d)) {  // behind the fp16 image: the Euclidean path carves the same layout and ignores the rest
        r.err = c.take<float>((size_t)rp);
        r.gstep = c.take<float>((size_t)rp / I8_GROUP);
        r.gerr = c.take<float>((size_t)rp / I8_GROUP);
        r.tiles8 = c.take<uint4>((size_t)rp / TILE_ROWS * (size_t)(d / 32) * 64);
    }
    r.bytes = c.used();
    return r;
}

struct SearchWs {
    uint2* partials;
    int* cand_cnt;
    unsigned* cand;
    int cap;  // entries per query in `cand`
    unsigned* rec_cnt;  // sparse records of the coarse pass (see CoarseArgs)
    uint2* rec;
    int rcap;
    int* fb_count;
    int* fb_list;
    unsigned* qmax;
    size_t bytes;
};

inline SearchWs carve_search(void* p, int64_t n, int64_t m) {
    VfmCarver c(p);
    SearchWs w;
    const int64_t npad = rows_padded(n), mpad = rows_padded(m);
    w.partials = c.take<uint2>((size_t)(mpad / CHUNK_ROWS) * (size_t)npad);
    w.cand_cnt = c.take<int>((size_t)npad);
    w.cap = cand_cap(mpad);
    w.cand = c.take<unsigned>((size_t)npad * (size_t)w.cap);
    w.fb_list = c.take<int>((size_t)npad);
    // zeroed before every search by ONE memset: [fb_count (64, padded to 256 B) | qmax (npad) | rec_cnt (npad)]; npad is a
    // multiple of 256, so the three arrays are contiguous under the carver's 256-byte alignment
    w.fb_count = c.take<int>(64);
    w.qmax = c.take<unsigned>((size_t)npad);
    w.rec_cnt = c.take<unsigned>((size_t)npad);
    w.rcap = FILTER_LDS_ROWS;
    w.rec = c.take<uint2>((size_t)npad * (size_t)w.rcap);
    w.bytes = c.used();
    return w;
}

int g_force_slices = 0;  // experiment knob (vfm_debug_set_coarse_slices): 0 = heuristic below

inline int choose_slices(int nqb, int nchunks) {
    if (g_force_slices > 0) return g_force_slices < nchunks ? g_force_slices : nchunks;
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
int g_coarse_qsets = 0;
float g_window_override = 0.0f;  // vfm_debug_set_coarse_window: timing experiments only (results are wrong)
int g_seed_units = 1;   // vfm_debug_set_coarse_variant(7): no seed units (A/B)
int g_match_stats = 0;  // vfm_debug_set_match_stats: per-query counters cost ~0.5 ms of same-address atomics per search
// 4 = pipelined kernel with the DENSE per-chunk records + match_select_kernel (round-1 path; A/B reference)

// inner-product searches with d <= 384 use the sparse row-level records of match_coarse_pipe_kernel<., true>
// (from 3 query blocks on: with 1-2 blocks every unit of the grid runs at once, none is seeded, and the dense records
// measured faster -- 244 vs 282 us per registration at 300 x 50000)
inline bool use_sparse(int d, int64_t n, int64_t m) {
    return d <= 384 && d % 128 == 0 && m < (1ll << 24) && n > 2 * QBLOCK &&
           (g_coarse_qsets == 0 || g_coarse_qsets == 3 || g_coarse_qsets == 5);
}

// ... and before those, in the GATED family of entry points (callers that keep only matches above a similarity gate), for
// d = 256 / 384: the int8 coarse pass with one best-score record per (query, chunk).  Its exact re-decision costs an int8
// rescan per candidate chunk, cheap when most unmatched queries stop at the gate and slower than the fp16 pass when every
// query must be resolved -- so the ungated entry points keep the fp16 pass.  (variant 5 = fp16 pass everywhere, A/B)
int g_i8_min_queries = 0;  // vfm_debug_set_i8_min_queries (A/B knob): the int8 pass wins at every size measured (300 x 50 000: 209 vs 244 us)
inline bool use_i8(int d, int64_t n, int64_t m, bool gated) {
    return gated && i8_capable(d) && m < (1ll << 24) && n > g_i8_min_queries &&
           (g_coarse_qsets == 0 || g_coarse_qsets == 10 || g_coarse_qsets == 12);
}
inline I8Bounds i8_bounds(const Prepared& Q, const Prepared& B, bool on, int top2 = 0) {
    return on ? I8Bounds{Q.err, Q.gstep, B.gstep, B.gerr, top2} : I8Bounds{nullptr, nullptr, nullptr, nullptr, 0};
}

// queries per workgroup of the coarse kernel that do_search_coarse will launch
int coarse_qblock(int d) { return d > 512 ? 128 : QBLOCK; }

// hipFuncSetAttribute is per device: remember which devices have been configured (one bit each)
inline bool attr_done(unsigned long long mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return (mask >> (dev & 63)) & 1ull;
}
inline void attr_mark(unsigned long long& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    mask |= 1ull << (dev & 63);
}

template <int KSTEPS, bool SPARSE>
int launch_coarse_pipe(const CoarseArgs& a, hipStream_t st) {
    const int lds = 6 * KSTEPS * 1024 + (SPARSE ? SPARSE_LREC_CAP * 8 + 16 : 0);
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_coarse_pipe_kernel<KSTEPS, SPARSE>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_mark(attr_set);
    }
    hipLaunchKernelGGL((match_coarse_pipe_kernel<KSTEPS, SPARSE>), dim3(a.nseed_pad + a.nqb * a.nslices), dim3(512), lds, st, a);
    return VFM_OK;
}

template <int KSTEPS, int T, bool TOP2 = false>
int launch_coarse_i8(const CoarseArgs& a, hipStream_t st) {
    const int lds = 3 * T * KSTEPS * 1024;
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_coarse_i8_kernel<KSTEPS, T, TOP2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_mark(attr_set);
    }
    hipLaunchKernelGGL((match_coarse_i8_kernel<KSTEPS, T, TOP2>), dim3(a.nqb * a.nslices), dim3(512), lds, st, a);
    return VFM_OK;
}

template <int KSTEPS, bool TOP2>
int launch_coarse_i8q2(const CoarseArgs& a, hipStream_t st) {
    const int lds = 12 * KSTEPS * 1024;
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_coarse_i8q2_kernel<KSTEPS, TOP2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_mark(attr_set);
    }
    hipLaunchKernelGGL((match_coarse_i8q2_kernel<KSTEPS, TOP2>), dim3(a.nqb * a.nslices), dim3(512), lds, st, a);
    return VFM_OK;
}

template <int KSTEPS, int QSETS>
int launch_coarse_v(const CoarseArgs& a, hipStream_t st) {
    const int lds = ring_depth(KSTEPS) * KSTEPS * 1024;
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_coarse_kernel<KSTEPS, QSETS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_mark(attr_set);
    }
    hipLaunchKernelGGL((match_coarse_kernel<KSTEPS, QSETS>), dim3(a.nqb * a.nslices), dim3(512 / QSETS), lds, st, a);
    return VFM_OK;
}

template <int KSTEPS, int QSETS, int NBUF, bool BIAS>
int launch_coarse_r(const CoarseArgs& a, hipStream_t st) {
    const int lds = NBUF * KSTEPS * 1024;
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_coarse_r_kernel<KSTEPS, QSETS, NBUF, BIAS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_mark(attr_set);
    }
    if (g_prof_start) VFM_CHECK_HIP(hipEventRecord(g_prof_start, st));
    hipLaunchKernelGGL((match_coarse_r_kernel<KSTEPS, QSETS, NBUF, BIAS>), dim3(a.nqb * a.nslices), dim3(256), lds, st, a);
    VFM_CHECK_LAUNCH("match_coarse_r_kernel");
    if (g_prof_stop) VFM_CHECK_HIP(hipEventRecord(g_prof_stop, st));
    g_prof_start = g_prof_stop = nullptr;
    return VFM_OK;
}

template <int KSTEPS>
int launch_coarse(const CoarseArgs& a, hipStream_t st) {
    if (g_prof_start) VFM_CHECK_HIP(hipEventRecord(g_prof_start, st));
    int rc;
    if constexpr (KSTEPS <= 24) {
        rc = (g_coarse_qsets == 2)   ? launch_coarse_v<KSTEPS, 2>(a, st)
             : (g_coarse_qsets == 1) ? launch_coarse_v<KSTEPS, 1>(a, st)
                                     : (a.rec ? launch_coarse_pipe<KSTEPS, true>(a, st) : launch_coarse_pipe<KSTEPS, false>(a, st));  // 0, 3
    } else {
        rc = launch_coarse_v<KSTEPS, 1>(a, st);  // d = 512: 2 x 128 query VGPRs would not fit
    }
    if (rc) return rc;
    VFM_CHECK_LAUNCH("match_coarse_kernel");
    if (g_prof_stop) VFM_CHECK_HIP(hipEventRecord(g_prof_stop, st));
    g_prof_start = g_prof_stop = nullptr;
    return VFM_OK;
}

inline PrepOut prep_out(const Prepared& p) { return PrepOut{p.inv, p.tiles, p.err, p.gstep, p.gerr, p.tiles8}; }

// one or two operands (x2 may be NULL) in one launch.  want_f16 = false: only the int8 image (d = 256, 384), for operands that
// will meet in an int8 search (use_i8): a third of the bytes written, a third of the LDS.
int do_prepare2(const float* x1, int64_t rows1, void* prepared1, const float* x2, int64_t rows2, void* prepared2, int d,
                hipStream_t st, bool want_f16 = true) {
    Prepared p1 = carve_prepared(prepared1, rows1, d);
    Prepared p2 = x2 ? carve_prepared(prepared2, rows2, d) : Prepared{};
    const int t1 = (int)(rows_padded(rows1) / TILE_ROWS), t2 = x2 ? (int)(rows_padded(rows2) / TILE_ROWS) : 0;
    if (i8_capable(d)) {  // int8 tiles + group data (+ fp16 tiles)
        const int g1 = (int)(rows_padded(rows1) / I8_GROUP), g2 = x2 ? (int)(rows_padded(rows2) / I8_GROUP) : 0;
        static unsigned long long attr_set = 0ull;  // one bit per device
        if (!attr_done(attr_set)) {
            VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prep_chunk_kernel<true, 2>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, I8_GROUP * 384 * 3));
            VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prep_chunk_kernel<false, 2>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, I8_GROUP * 512));
            VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prep_chunk_kernel<false, 3>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, I8_GROUP * 768));
            attr_mark(attr_set);
        }
        const dim3 grid((unsigned)(g1 + g2)), block(1024);
        if (want_f16 && d <= 384) {  // both images from one read of the rows (144 KB of LDS at d = 384)
            hipLaunchKernelGGL((prep_chunk_kernel<true, 2>), grid, block, (size_t)I8_GROUP * d * 3, st, x1, rows1, d, prep_out(p1), g1, x2,
                               rows2, prep_out(p2));
        } else {
            if (d <= 512)
                hipLaunchKernelGGL((prep_chunk_kernel<false, 2>), grid, block, (size_t)I8_GROUP * d, st, x1, rows1, d, prep_out(p1), g1, x2,
                                   rows2, prep_out(p2));
            else
                hipLaunchKernelGGL((prep_chunk_kernel<false, 3>), grid, block, (size_t)I8_GROUP * d, st, x1, rows1, d, prep_out(p1), g1, x2,
                                   rows2, prep_out(p2));
            if (want_f16) {  // wider rows: the fp16 image by its own kernel (both images would not fit the LDS)
                hipLaunchKernelGGL(prep_rows_kernel, dim3((unsigned)(t1 + t2)), dim3(256), (size_t)d * 64, st, x1, rows1, d, p1.inv,
                                   p1.tiles, t1, x2, rows2, p2.inv, p2.tiles);
            }
        }
        VFM_CHECK_LAUNCH("prep_chunk_kernel");
        return VFM_OK;
    }
    hipLaunchKernelGGL(prep_rows_kernel, dim3((unsigned)(t1 + t2)), dim3(256), (size_t)d * 64, st, x1, rows1, d, p1.inv, p1.tiles,
                       t1, x2, rows2, p2.inv, p2.tiles);
    VFM_CHECK_LAUNCH("prep_rows_kernel");
    return VFM_OK;
}

int do_prepare(const float* x, int64_t rows, int d, void* prepared, hipStream_t st) {
    return do_prepare2(x, rows, prepared, nullptr, 0, nullptr, d, st);
}

CoarseArgs coarse_args(const Prepared& Q, const Prepared& B, const SearchWs& w, int64_t n, int64_t m, int qblock = QBLOCK) {
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
    a.window = g_window_override != 0.0f ? g_window_override : DEFAULT_WINDOW;
    a.ib = I8Bounds{nullptr, nullptr, nullptr, nullptr, 0};
    return a;
}

// stage 1 of a search: the MFMA coarse pass (fills ws: partials + per-query coarse maxima)
// records (int8 pass): 0 = best score per (query, chunk), 1 = packed top-2 with the best row's index (VFM_RECORDS_*)
int do_search_coarse(const void* qprep, int64_t n, const void* bprep, int64_t m, int d, void* ws, hipStream_t st,
                     bool bias_from_map_inv = false, bool inner_product = false, bool gated = false, int records = 0) {
    Prepared Q = carve_prepared(const_cast<void*>(qprep), n, d);
    Prepared B = carve_prepared(const_cast<void*>(bprep), m, d);
    SearchWs w = carve_search(ws, n, m);
    CoarseArgs a = coarse_args(Q, B, w, n, m, coarse_qblock(d));
    if (bias_from_map_inv) {  // Euclidean search, d > 510: the map's "inv" array holds -|b~|^2 / 2
        if (d != 640 && d != 768) return vfm_fail(VFM_EINVAL, "row bias needs the 4-wave coarse kernel (K = 640 / 768), got %d", d);
        a.row_bias = B.inv;
    }
    if (inner_product && !use_i8(d, n, m, gated) && use_sparse(d, n, m)) {
        a.rec_cnt = w.rec_cnt;
        a.rec = w.rec;
        a.rcap = w.rcap;
        if (g_seed_units && a.nchunks >= 256 && a.nqb <= 256) {  // seed units: one short round at the head of the grid
            a.seed_parts = 256 / a.nqb < 4 ? 256 / a.nqb : 4;
            a.seed_chunks = 5;
            a.nseed_pad = (a.nqb * a.seed_parts + 7) / 8 * 8;
            a.nslices = choose_slices(a.nqb, a.nchunks - a.seed_parts * a.seed_chunks);
        }
    }
    VFM_CHECK_HIP(hipMemsetAsync(w.fb_count, 0, 256 + 2 * (size_t)a.npad * sizeof(unsigned), st));  // fb_count | qmax | rec_cnt
    if (inner_product && use_i8(d, n, m, gated)) {
        a.Qh = Q.tiles8;
        a.Bh = B.tiles8;
        a.ib = I8Bounds{Q.err, Q.gstep, B.gstep, B.gerr, records};
        if (g_prof_start) VFM_CHECK_HIP(hipEventRecord(g_prof_start, st));
        a.nqb = (int)(rows_padded(n) / QBLOCK);  // 8 waves x 32 queries at every width
        a.nslices = choose_slices(a.nqb, a.nchunks);
        int rc8;
        const bool top2 = records != 0;
        if (d <= 384 && n > 2048 && g_coarse_qsets == 0) {
            // 64 resident queries per wave: 11-15 % faster than the one-set kernel from ~3000 queries on (C2: 1.09 vs 1.23 ms;
            // 1500 x 100 000: 0.059 vs 0.056 ms -- half as many, twice as large workgroups); variants 10 / 12 = one set, A/B
            a.nqb = (a.nq_tiles + 15) / 16;
            a.nslices = choose_slices(a.nqb, a.nchunks);
            rc8 = d == 384 ? (top2 ? launch_coarse_i8q2<12, true>(a, st) : launch_coarse_i8q2<12, false>(a, st))
                           : (top2 ? launch_coarse_i8q2<8, true>(a, st) : launch_coarse_i8q2<8, false>(a, st));
        } else {
            const bool t2 = g_coarse_qsets == 10;  // variant 10 (A/B): 2 tiles per step at every width
            switch (d / 32) {
                case 8: rc8 = top2 ? launch_coarse_i8<8, 4, true>(a, st) : t2 ? launch_coarse_i8<8, 2>(a, st) : launch_coarse_i8<8, 4>(a, st); break;
                case 12: rc8 = top2 ? launch_coarse_i8<12, 4, true>(a, st) : t2 ? launch_coarse_i8<12, 2>(a, st) : launch_coarse_i8<12, 4>(a, st); break;
                case 16: rc8 = top2 ? launch_coarse_i8<16, 2, true>(a, st) : launch_coarse_i8<16, 2>(a, st); break;
                case 20: rc8 = top2 ? launch_coarse_i8<20, 2, true>(a, st) : launch_coarse_i8<20, 2>(a, st); break;
                default: rc8 = top2 ? launch_coarse_i8<24, 2, true>(a, st) : launch_coarse_i8<24, 2>(a, st); break;
            }
        }
        if (rc8) return rc8;
        VFM_CHECK_LAUNCH("match_coarse_i8_kernel");
        if (g_prof_stop) VFM_CHECK_HIP(hipEventRecord(g_prof_stop, st));
        g_prof_start = g_prof_stop = nullptr;
        return VFM_OK;
    }
    int rc;
    switch (d / 16) {
        case 8: rc = launch_coarse<8>(a, st); break;
        case 16: rc = launch_coarse<16>(a, st); break;
        case 24: rc = launch_coarse<24>(a, st); break;
        case 32: rc = launch_coarse<32>(a, st); break;  // 8-wave kernel, ring of 4: 3.45 ms at C2 x 512 (4-wave kernel: 4.42 ms)
        case 40: rc = a.row_bias ? launch_coarse_r<40, 1, 3, true>(a, st) : launch_coarse_r<40, 1, 3, false>(a, st); break;
        case 48: rc = a.row_bias ? launch_coarse_r<48, 1, 3, true>(a, st) : launch_coarse_r<48, 1, 3, false>(a, st); break;
        default: return vfm_fail(VFM_EINVAL, "FAST matching supports d in {128,256,384,512,640,768}, got %d", d);
    }
    return rc;
}

// stage 2 of a search: candidate selection + exact fp64 decision (reads ws of stage 1)
// gated: the search was started by the gated family (do_search_coarse(..., gated)); gate: queries whose best similarity is
// provably below it are reported as (-1, -2.0) instead of being resolved (int8 pass only; -Inf = resolve every query)
int do_search_finish(const float* q, const void* qprep, int64_t n, const float* b, const void* bprep, int64_t m, int d,
                     int64_t* idx_out, float* sim_out, void* ws, hipStream_t st, bool gated = false,
                     float gate = -__builtin_inff(), int records = 0) {
    Prepared Q = carve_prepared(const_cast<void*>(qprep), n, d);
    Prepared B = carve_prepared(const_cast<void*>(bprep), m, d);
    SearchWs w = carve_search(ws, n, m);
    const CoarseArgs a = coarse_args(Q, B, w, n, m, coarse_qblock(d));
    const float w2 = 2.0f * (float)(d / 16 + 4 + 2) * 5.9604645e-8f;
    const bool i8 = use_i8(d, n, m, gated);
    if (!i8 && use_sparse(d, n, m)) {
        hipLaunchKernelGGL(match_filter_refine_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, q, Q.inv, b, B.inv, n, m, d,
                           DEFAULT_WINDOW, w2, w.qmax, w.rec_cnt, w.rec, w.rcap, w.cand_cnt, w.cand, w.cap, w.fb_count, w.fb_list, g_match_stats);
        VFM_CHECK_LAUNCH("match_filter_refine_kernel");
    } else {
        const int chunk_lds = i8 && (size_t)a.nchunks * sizeof(float2) <= 63 * 1024;  // (step, max E) of every chunk in LDS
        hipLaunchKernelGGL(match_select_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * SELECT_GROUPS),
                           chunk_lds ? (size_t)a.nchunks * sizeof(float2) : 0, st, w.partials, a.nchunks, a.npad, n, a.first_pad_chunk, w.qmax,
                           Q.inv, DEFAULT_WINDOW, i8_bounds(Q, B, i8, records), gate, chunk_lds, w.cand_cnt, w.cand, w.cap, w.fb_count, w.fb_list,
                           g_match_stats);
        VFM_CHECK_LAUNCH("match_select_kernel");
        if (i8) {  // candidate chunks -> candidate rows (the record buffer of the fp16 pass is free: it holds the hit lists)
            hipLaunchKernelGGL(match_rescan_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, m, d, i8_bounds(Q, B, true, records),
                               (const uint4*)Q.tiles8, (const uint4*)B.tiles8, (const unsigned*)w.qmax, w.cand_cnt, w.cand, w.cap,
                               reinterpret_cast<unsigned*>(w.rec), 2 * w.rcap, w.fb_count, w.fb_list,
                               reinterpret_cast<int*>(w.rec_cnt));
            VFM_CHECK_LAUNCH("match_rescan_kernel");
            // (rec_cnt, unused by the int8 pass, holds the list of crowded queries; fb_count[6] its length)
            const unsigned grid = (unsigned)((n + 3) / 4 < 1024 ? (n + 3) / 4 : 1024);
            hipLaunchKernelGGL(match_refine_kernel, dim3(grid), dim3(256), 0, st, q, Q.inv, b, B.inv, n, m, d, w2, w.cand_cnt, w.cand,
                               w.cap, w.fb_count, w.fb_list, g_match_stats, (const int*)w.rec_cnt, (const int*)(w.fb_count + 6));
        } else {
            hipLaunchKernelGGL(match_refine_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, q, Q.inv, b, B.inv, n, m, d, w2,
                               w.cand_cnt, w.cand, w.cap, w.fb_count, w.fb_list, g_match_stats, (const int*)nullptr, (const int*)nullptr);
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
                           d, w.cand_cnt, w.cand, w.cap, idx_out, sim_out);
    }
    VFM_CHECK_LAUNCH("match_rescore_kernel");
    hipLaunchKernelGGL(match_exact_kernel, dim3(256), dim3(256), (((size_t)d * 4 + 15) & ~(size_t)15) + 64, st, q, Q.inv,
                       b, B.inv, n, m, d, w.fb_list, w.fb_count, idx_out, sim_out);
    VFM_CHECK_LAUNCH("match_exact_kernel(fallback)");
    return VFM_OK;
}

int do_search(const float* q, const void* qprep, int64_t n, const float* b, const void* bprep, int64_t m, int d,
              int64_t* idx_out, float* sim_out, void* ws, hipStream_t st) {
    const int rc = do_search_coarse(qprep, n, bprep, m, d, ws, st, false, true);
    if (rc) return rc;
    return do_search_finish(q, qprep, n, b, bprep, m, d, idx_out, sim_out, ws, st);
}

}  // namespace

VFM_EXPORT int vfm_l2norm_rows_f32(float* x, int64_t n, int d, float* inv_out, vfm_stream_t stream) {
    VFM_CHECK_ARG(n >= 0 && d > 0 && d % 4 == 0 && d <= 1024, "l2norm: need d %% 4 == 0 and d <= 1024 (d=%d)", d);
    if (n == 0) return VFM_OK;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, n, d,
                       inv_out);
    VFM_CHECK_LAUNCH("l2norm_rows_kernel");
    return VFM_OK;
}

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
    VFM_CHECK_ARG(records == VFM_RECORDS_BEST || records == VFM_RECORDS_TOP2, "search_coarse: unknown record kind %d", records);
    return do_search_coarse(q_prepared, n, b_prepared, m, d, ws, (hipStream_t)stream, false, true, true, records);
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
    VFM_CHECK_ARG(records == VFM_RECORDS_BEST || records == VFM_RECORDS_TOP2, "search_finish: unknown record kind %d", records);
    return do_search_finish(q, q_prepared, n, b, b_prepared, m, d, idx_out, sim_out, ws, (hipStream_t)stream, true, gate, records);
}

VFM_EXPORT int vfm_match_search_rescans_async(const void* ws, int64_t n, int64_t m, int32_t* out_host, vfm_stream_t stream) {
    VFM_CHECK_ARG(ws && out_host && n > 0 && m > 0, "search_rescans: bad arguments");
    SearchWs w = carve_search(const_cast<void*>(ws), n, m);
    VFM_CHECK_HIP(hipMemcpyAsync(out_host, w.fb_count + 5, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return VFM_OK;
}

VFM_EXPORT size_t vfm_match_ip_top1_workspace_bytes(int64_t n, int64_t m, int d, int prec_mode) {
    if (prec_mode == VFM_MATCH_EXACT) return vfm_align_up((size_t)(n + m) * sizeof(float), 256) + 512;
    return vfm_match_prepared_bytes(n, d) + vfm_match_prepared_bytes(m, d) + vfm_match_search_workspace_bytes(n, m, d);
}

namespace {
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

namespace {
int ip_top1(const float* q, int64_t n, const float* b, int64_t m, int d, int prec_mode, bool gated, float gate, int64_t* idx_out,
            float* sim_out, void* ws, size_t ws_bytes, vfm_stream_t stream);
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
namespace {
int ip_top1(const float* q, int64_t n, const float* b, int64_t m, int d, int prec_mode, bool gated, float gate, int64_t* idx_out,
            float* sim_out, void* ws, size_t ws_bytes, vfm_stream_t stream) {
    VFM_CHECK_ARG(n > 0 && m > 0, "match: empty operand (n=%lld m=%lld)", (long long)n, (long long)m);
    VFM_CHECK_ARG(q && b && idx_out && sim_out && ws, "match: null pointer");
    if (ws_bytes < vfm_match_ip_top1_workspace_bytes(n, m, d, prec_mode)) return vfm_fail(VFM_EWORKSPACE, "match: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (prec_mode == VFM_MATCH_EXACT) {
        VFM_CHECK_ARG(d % 4 == 0 && d <= 1024, "match(EXACT): need d %% 4 == 0 and d <= 1024");
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
}  // namespace

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

namespace {
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
    hipLaunchKernelGGL(match_select_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * SELECT_GROUPS), 0, st, w.partials, a.nchunks,
                       a.npad, n, a.first_pad_chunk, w.qmax, Q.inv, DEFAULT_WINDOW, i8_bounds(Q, Q, false), -__builtin_inff(), 0, w.cand_cnt, w.cand,
                       w.cap, w.fb_count, w.fb_list, g_match_stats);
    VFM_CHECK_LAUNCH("match_select_kernel");
    hipLaunchKernelGGL(l2_rescore_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)d * 4 * 8, st, q, b, n, m, d, w.cand_cnt,
                       w.cand, w.cap, nn, d2);
    VFM_CHECK_LAUNCH("l2_rescore_kernel");
    const size_t lds = (((size_t)d * 4 + 15) & ~(size_t)15) + 64;
    hipLaunchKernelGGL(nn_l2_kernel, dim3(256), dim3(256), lds, st, q, n, b, m, d, w.fb_list, w.fb_count, nn, d2);
    VFM_CHECK_LAUNCH("nn_l2_kernel(fallback)");
    return VFM_OK;
}
}  // namespace

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

VFM_EXPORT int vfm_debug_set_i8_min_queries(int n) {
    g_i8_min_queries = n;
    return VFM_OK;
}

VFM_EXPORT int vfm_debug_set_coarse_window(float w) {
    g_window_override = w;
    return VFM_OK;
}

VFM_EXPORT int vfm_debug_set_match_stats(int on) {
    g_match_stats = on;
    return VFM_OK;
}

VFM_EXPORT int vfm_debug_set_coarse_slices(int slices) {
    g_force_slices = slices;
    return VFM_OK;
}
VFM_EXPORT int vfm_debug_set_coarse_variant(int qsets) {
    g_seed_units = qsets == 7 ? 0 : 1;
    if (qsets == 7) qsets = 0;
    g_coarse_qsets = qsets;
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
