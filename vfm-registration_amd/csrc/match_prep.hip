// match_prep.hip -- operand preparation of the matcher (VoxelHashMap.cpp:469-482): 1/|row| in faiss' fvec_renorm_L2 order,
// the fp16 MFMA-fragment image of the normalised rows (prep_rows_kernel) and the int8 image with per-128-row quantisation
// steps and measured residual norms (prep_chunk_kernel).  Compiled with -ffp-contract=off.
#include "match_internal.h"

namespace vfmm {
namespace {

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
    uint4* tiles8h;   // int8 fragment tiles of the first d / 2 columns
    float* rest;      // |second half of the normalised row|_2, rounded up
    float* grest;     // its maximum over the group
    const int* perm;  // row r of the image is row perm[r] of x (NULL: identity)
    uint4* tiles6;    // MX6: fp6 fragment tiles
    float* err6;      // MX6: residual norm of the fp6 image per row (+ MX6_SLACK)
    float* gerr6;     // MX6: its maximum over the group
    float* gstep6;    // MX6: MX6_FIX_STEP
    float* err6h;     // MX6: residual norm of the fp6 image over the first d / 2 columns (+ MX6_SLACK)
    float* gerr6h;    // MX6: its maximum over the group
    unsigned char* rows8;   // the int8 image row-major (NULL: operand too large to be a scan, Prepared::rows8)
    int mx6_half;     // MX6: VFM_PREPARE_MX6_HALF -- only the first d / 2 columns are converted (the tile prefix the half-width pass reads), err6 /
                      // gerr6 are infinite, no int8 half-width image is written
};
// MX6: additionally the microscaled fp6 image (OCP MX: e2m3 elements, one power-of-two scale per 32 columns) in fragment tiles
// of v_mfma_scale_f32_32x32x64_f8f6f4.  Lane l of that MFMA holds, for row l & 31, the 32 columns 64 s + 32 (l >> 5) ... of
// k-step s as 32 consecutive 6-bit codes (little-endian: code f at bits [6 f, 6 f + 6) -- tools/probe/mx6_probe.hip checks
// the layout on the device), i.e. exactly ONE scale block.  The stored tile is dense (match_internal.h, mx6_tile_bytes / mx6_code_a /
// mx6_code_b / mx6_scale_at): a plane of the lanes' scales -- 8 bytes per lane, byte s = the E8M0 scale of k-step s: one 8-byte
// read per tile, the MFMA's op_sel picks the byte --, then per k-step a plane of the lanes' first 16 code bytes and a plane of
// their last 8.
// The image is made from the fp16 copy of the normalised rows that phase 2 leaves in the LDS (the staging of the fp16 image):
// once the int8 tiles have left the LDS, a thread takes one (row, 32-column block) -- 1536 of them per group -- reads its 32
// halves, picks the scale 2^e with max / 2^e <= 7.75 (the largest code is 7.5; up to 7.75 rounds there with the half-step
// error of its binade) and converts with v_cvt_scalef32_pk32_fp6_f16 (round to nearest even, saturating:
// tools/probe/mx6_cvt_probe.hip checks it against the arithmetic definition); the residual against the fp16 values is MEASURED
// (codes converted back by v_cvt_scalef32_pk32_f16_fp6 at scale 1: exact), the rounding of the fp16 copy itself is bounded:
// |v - fp16(v)|_2 <= 2^-11 |v|_2 + sqrt(d) 2^-25 (denormals), |v|_2 <= 1 + 2^-13.
constexpr float MX6_SLACK = 4.0e-5f;   // on every E: the MFMA's fp32 accumulation (<= 6 steps x a few ulp of 4), the records' 2^-20 grid, the top-2 packing (64 ulp of 4)
constexpr float PREP_F16_ROUNDING = 4.8929e-4f;  // the same bound, for the int8 image prep_once_kernel quantises from the fp16 copy
constexpr float MX6_F16_ROUNDING = 4.8929e-4f;   // 2^-11 (1 + 2^-13) + sqrt(768) 2^-25, rounded up
typedef _Float16 halfx32 __attribute__((ext_vector_type(32)));
typedef int intx6 __attribute__((ext_vector_type(6)));
typedef unsigned short ushortx2 __attribute__((ext_vector_type(2)));
// WAVES = 8 (the MX6 form): 8 waves x 16 rows -- half the threads, twice the registers each: room for the conversion AND for the
// next group's rows, which the 16-wave form had to read after it (128 registers: 49 spilled, the loads exposed; 0.21 ms)
template <bool F16, int NC = 2, bool MX6 = false, int WAVES = 16>
__global__ __launch_bounds__(WAVES * 64) void prep_chunk_kernel(Rows x1, int64_t rows1, int d, PrepOut o1, int groups1,
                                                          Rows x2, int64_t rows2, PrepOut o2, int groups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned amax_bits, emax_bits, rmax_bits, e6max_bits, e6hmax_bits;
    static_assert(!(F16 && MX6), "the fp6 image is made from the LDS copy of the fp16 one, which then is not stored");
    constexpr int RPW = I8_GROUP / WAVES;  // rows per wave (8 or 16), all of them in registers between the two phases
    constexpr int NT = WAVES * 64;
    static_assert(RPW % 8 == 0, "rows are reduced eight at a time");
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int nchunks = d >> 2;  // float4 chunks per row (<= 64 NC): lane l owns chunks l, l + 64 (, l + 128)
    unsigned char* img8 = smem;                                                  // [4 tiles][d/32 * 64 units][16]
    _Float16* img16 = reinterpret_cast<_Float16*>(smem + (size_t)I8_GROUP * d);  // F16: [4 tiles][d/16 * 64 units][8]
    unsigned char* img6 = smem;                                                  // MX6: [4 tiles][d/64 * 128 units][16], over img8 once that is stored
    float* e6p = reinterpret_cast<float*>(smem + (size_t)I8_GROUP * d * 3);      // MX6: [d/32 blocks][128 rows] squared residuals
    unsigned char* scb = reinterpret_cast<unsigned char*>(e6p + (d >> 5) * I8_GROUP);   // MX6: [128 rows][16] block scales (E8M0)
    if (threadIdx.x == 0) {
        amax_bits = 0u;
        emax_bits = 0u;
        rmax_bits = 0u;
        e6max_bits = 0u;
        e6hmax_bits = 0u;
    }
    // The kernel's registers allow one workgroup per compute unit, so a workgroup walks several groups (grid = compute
    // units) and reads row j of its NEXT group as soon as row j of the current one has been quantised: the read of the next
    // 196 KB runs under the quantisation, the LDS transpose and the store of the current group instead of after them.
    // (a second operand -- x2, rows2, o2 -- rides in the same grid: groups >= groups1 are its groups)
    float4 v[RPW][NC];
    auto load_row = [&](int grp, int j) __attribute__((always_inline)) {
        const bool second = grp >= groups1;
        const Rows x = second ? x2 : x1;
        const int64_t rows = second ? rows2 : rows1;
        const int64_t r = (int64_t)(second ? grp - groups1 : grp) * I8_GROUP + wave * RPW + j;
        const int* perm = second ? o2.perm : o1.perm;
        const int64_t rsrc = (perm && r < rows) ? (int64_t)perm[r] : r;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef VFM_PABL_NOLOAD   // (timing experiments, tools/ablate_prep.py: results are garbage)
            if (r < rows && c < nchunks) t = make_float4(0.01f * (float)(c & 15), 0.5f, -0.25f, 1.0f);
            if (false) {
#else
            if (r < rows && c < nchunks) {
#endif
                t = x.ld4_nt(rsrc * (int64_t)d + 4 * c);   // (fp16 rows: widened here, VFM_ROWS_F16)
            }
            v[j][i] = t;
        }
    };
    // Eight per-row sums per lane -> one per lane: a reduce-scatter over the xor-32 / 16 / 8 levels (the lane keeps half of
    // its rows at every level and adds the partner's partial of those rows), then the xor-4 / 2 / 1 levels on the single
    // value.  Every addition pairs the same two partials as row_sumsq_wave's butterfly (which computes each of them in both
    // lanes), so the sum is bit-identical to the oracle's; 10 shuffles instead of 48.  Afterwards lane l holds row l >> 3.
    auto scatter8 = [&](const float* p) __attribute__((always_inline)) {
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
    if ((int)blockIdx.x < groups) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) load_row(blockIdx.x, j);
    }
    for (int gidx = blockIdx.x; gidx < groups; gidx += gridDim.x) {
        const bool second = gidx >= groups1;
        const int grp = second ? gidx - groups1 : gidx;
        const int64_t rows = second ? rows2 : rows1;
        const PrepOut& o = second ? o2 : o1;
        const int gnext = gidx + gridDim.x;
        // phase 1: 1/|row| in the oracle's order, the group's largest normalised magnitude
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
        float my_inv[RPW / 8];   // of row 8 hb + (lane >> 3): eight rows in one evaluation
#pragma unroll
        for (int hb = 0; hb < RPW / 8; ++hb) {
            my_inv[hb] = inv_norm_from_sumsq(scatter8(part + 8 * hb));
            if ((lane & 7) == 0) o.inv[(int64_t)grp * I8_GROUP + wave * RPW + 8 * hb + (lane >> 3)] = my_inv[hb];
        }
        float lmax = 0.0f;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const float iv = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_inv[j >> 3]), 8 * (j & 7)));
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
        __syncthreads();  // amax_bits / emax_bits initialised; the previous group's image has left the LDS
        if (lane == 0 && lmax > 0.0f) atomicMax(&amax_bits, __float_as_uint(lmax));  // finite or +Inf: uint order == float order
        __syncthreads();
        const float amax = __uint_as_float(amax_bits);
        const bool usable = amax > 0.0f && amax < 3.0e38f;
        const float qstep = usable ? amax / 127.0f : 1.0f;
        const float inv_qstep = usable ? 127.0f / amax : 0.0f;
        // phase 2: quantise from the registers; each row's registers then take the same row of the next group
        float rpart[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int pr = wave * RPW + j;
            const int t = pr >> 5, p = pr & 31;
            float e2 = 0.0f, r2 = 0.0f;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int c = lane + 64 * i;
#ifdef VFM_PABL_NOP2
                if (c < nchunks && v[j][i].x == 12345.0f) {
#else
                if (c < nchunks) {
#endif
                    const float nv[4] = {v[j][i].x, v[j][i].y, v[j][i].z, v[j][i].w};
                    if (8 * c >= d) r2 = r2 + (nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2] + nv[3] * nv[3]);  // columns >= d / 2
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
                    if constexpr (F16 || MX6) {
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
            rpart[j] = r2;
            if ((!MX6 || WAVES == 8) && gnext < groups) load_row(gnext, j);   // (MX6 with 16 waves: no registers left, rows are read below)
        }
#pragma unroll
        for (int hb = 0; hb < RPW / 8; ++hb) {
            // |e|_2 of row lane >> 3, rounded up: the fp32 sum of d non-negative terms is within (d + 8) 2^-24 of exact, sqrtf
            // within 2^-24
            float en = sqrtf(scatter8(part + 8 * hb)) * 1.000244140625f + 1.0e-30f;
            if (!(en == en)) en = __builtin_inff();
            const int64_t r = (int64_t)grp * I8_GROUP + wave * RPW + 8 * hb + (lane >> 3);
            if (r >= rows) en = 0.0f;
            if ((lane & 7) == 0) {
                o.err[r] = en;
                if (en > 0.0f) atomicMax(&emax_bits, __float_as_uint(en));
                // no fp6 image in this form: its E says so -- a search that asks for an fp6 record kind on such an operand bounds
                // nothing and ends in the exact all-pairs decision (correct, slow) instead of trusting stale bytes
                if constexpr (!MX6) {
                    if (o.err6) o.err6[r] = o.err6h[r] = __builtin_inff();
                }
            }
            {   // (the MX6 form keeps these: VFM_RECORDS_MX6_HALF bounds the second half of the columns with them)
                // |second half of the row|_2, rounded up like E (d / 2 + 8 roundings of 2^-24 on non-negative terms, one sqrtf);
                // NaN / Inf elements -> Inf: nothing is ever pruned against such a row
                float rn = sqrtf(scatter8(rpart + 8 * hb)) * 1.000244140625f + 1.0e-30f;
                if (!(rn == rn)) rn = __builtin_inff();
                if (r >= rows) rn = 0.0f;
                if ((lane & 7) == 0) {
                    o.rest[r] = rn;
                    if (rn > 0.0f) atomicMax(&rmax_bits, __float_as_uint(rn));
                }
            }
        }
        __syncthreads();
        {
            const int u8n = (d >> 5) * 64 * 4;  // uint4 units of the group's four int8 tiles
            uint4* dst = o.tiles8 + (int64_t)grp * u8n;
            const uint4* src = reinterpret_cast<const uint4*>(img8);
#ifdef VFM_PABL_NOST8
            for (int u = threadIdx.x; u < u8n && d < 0; u += NT) {
#else
            for (int u = threadIdx.x; u < u8n; u += NT) {
#endif
                const uint4 tq = src[u];
                unsigned* po = reinterpret_cast<unsigned*>(dst + u);
                __builtin_nontemporal_store(tq.x, po);
                __builtin_nontemporal_store(tq.y, po + 1);
                __builtin_nontemporal_store(tq.z, po + 2);
                __builtin_nontemporal_store(tq.w, po + 3);
            }
        }
        if (o.rows8) {   // ... and row-major (scan-sized operands): unit un of row `row` sits at tile row >> 5, column un, row & 31
            const int nu = d >> 4;
            const uint4* src = reinterpret_cast<const uint4*>(img8);
            for (int u = threadIdx.x; u < I8_GROUP * nu; u += NT) {
                const int row = u / nu, un = u - row * nu;
                const uint4 tq = src[(row >> 5) * (nu * 32) + un * 32 + (row & 31)];
                *reinterpret_cast<uint4*>(o.rows8 + ((size_t)grp * I8_GROUP + row) * (size_t)d + 16 * un) = tq;
            }
        }
        if (!(MX6 && o.mx6_half)) {  // the first d / 2 columns again, as tiles of their own (the first half of every tile's units)
            const int uh = (d >> 6) * 64;  // uint4 units per half tile
            uint4* dst = o.tiles8h + (int64_t)grp * (uh * 4);
            const uint4* src = reinterpret_cast<const uint4*>(img8);
            for (int u = threadIdx.x; u < uh * 4; u += NT) {
                const int t = u / uh, w = u % uh;
                const uint4 tq = src[t * (2 * uh) + w];
                unsigned* po = reinterpret_cast<unsigned*>(dst + u);
                __builtin_nontemporal_store(tq.x, po);
                __builtin_nontemporal_store(tq.y, po + 1);
                __builtin_nontemporal_store(tq.z, po + 2);
                __builtin_nontemporal_store(tq.w, po + 3);
            }
        }
#ifdef VFM_PABL_NOMX6
        if constexpr (false) {
#else
        if constexpr (MX6) {
#endif
            __syncthreads();   // the int8 tiles have left the LDS: the fp6 image takes their place
            const int nblk = d >> 5;
#ifdef VFM_PABL_NOCONV
            const int nconv = 0;
#else
            const int nconv = o.mx6_half ? nblk >> 1 : nblk;   // blocks converted: VFM_PREPARE_MX6_HALF stops at column d / 2
#endif
            const int tb6 = mx6_tile_bytes(d >> 6);   // (d <= 384 here: one scale plane)
            for (int item = threadIdx.x; item < I8_GROUP * nconv; item += NT) {
                const int r = item & (I8_GROUP - 1), blk = item >> 7, t = r >> 5, p = r & 31;
                // the block's 32 halves: fp16 units (k-step 2 blk + u, half hh) of row p, in column order
                const uint4* up = reinterpret_cast<const uint4*>(img16 + (size_t)t * (d * 32)) + (4 * blk) * 32 + p;
                union {
                    uint4 u[4];
                    halfx32 h;
                    unsigned w[16];
                } v;
                v.u[0] = up[0];
                v.u[1] = up[32];
                v.u[2] = up[64];
                v.u[3] = up[96];
                ushortx2 m2 = {0, 0};   // packed maximum of the magnitudes (as 15-bit integers: the order of non-negative halves)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned a2 = v.w[i] & 0x7fff7fffu;
                    m2 = __builtin_elementwise_max(m2, *reinterpret_cast<const ushortx2*>(&a2));
                }
                const unsigned am = max((unsigned)m2[0], (unsigned)m2[1]);
                // am = (1 + f) 2^x: e = x - 2 when 1 + f <= 1.9375 (max / 2^e <= 7.75), else x - 1; Inf / NaN blocks: scale 1 (the
                // residual turns E into Inf); normalised finite rows: e <= -2
                int ex = (int)(am >> 10) - 15 - ((am & 0x3ffu) <= 0x3c0u ? 2 : 1);
                ex = ex > 0 ? 0 : ex;
                const float sc6 = __uint_as_float((unsigned)(ex + 127) << 23);
                const intx6 codes = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v.h, sc6);
                const halfx32 back = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(codes, 1.0f);   // the code values: exact in fp16
                float e6 = 0.0f;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const float res = __builtin_fmaf(-(float)back[i], sc6, (float)v.h[i]);
                    e6 = __builtin_fmaf(res, res, e6);
                }
                const int s6 = blk >> 1, l6 = (blk & 1) * 32 + p;
                unsigned char* tile6 = img6 + (size_t)t * tb6;
                *reinterpret_cast<uint4*>(tile6 + mx6_code_a(s6, l6)) =
                    make_uint4((unsigned)codes[0], (unsigned)codes[1], (unsigned)codes[2], (unsigned)codes[3]);
                *reinterpret_cast<uint2*>(tile6 + mx6_code_b(s6, l6)) = make_uint2((unsigned)codes[4], (unsigned)codes[5]);
                scb[r * 16 + blk] = (unsigned char)(ex + 127);
                e6p[blk * I8_GROUP + r] = e6;
            }
            __syncthreads();
            if (threadIdx.x < 2 * I8_GROUP) {   // the d / 64 scales of MFMA lane (hh, p) of tile t: its 8 bytes of the scale plane
                const int r = threadIdx.x & (I8_GROUP - 1), hh = threadIdx.x >> 7;
                unsigned lo = 0u, hi = 0u;
                for (int s6 = 0; s6 < (nconv >> 1); ++s6) {
                    const unsigned b = scb[r * 16 + 2 * s6 + hh];
                    if (s6 < 4) lo |= b << (8 * s6);
                    else hi |= b << (8 * (s6 - 4));
                }
                *reinterpret_cast<uint2*>(img6 + (size_t)(r >> 5) * tb6 + mx6_scale_at(d >> 6, 0, hh * 32 + (r & 31))) = make_uint2(lo, hi);
            } else if (threadIdx.x < 3 * I8_GROUP) {   // E of the fp6 image per row: blocks in order; rounded up like the int8 one
                const int r = threadIdx.x - 2 * I8_GROUP;
                float acc = 0.0f, acch = 0.0f;
                for (int blk = 0; blk < nconv; ++blk) {
                    acc = acc + e6p[blk * I8_GROUP + r];
                    if (blk + 1 == (nblk >> 1)) acch = acc;   // the first d / 2 columns: the same additions in the same order
                }
                float e6n = sqrtf(acc) * 1.000244140625f + (MX6_F16_ROUNDING + MX6_SLACK);
                float e6h = sqrtf(acch) * 1.000244140625f + (MX6_F16_ROUNDING + MX6_SLACK);
                if (!(e6n == e6n) || o.mx6_half) e6n = __builtin_inff();   // (no full-width image in that form: its E says so)
                if (!(e6h == e6h)) e6h = __builtin_inff();
                const int64_t row = (int64_t)grp * I8_GROUP + r;
                if (row >= rows) e6n = e6h = 0.0f;
                o.err6[row] = e6n;
                o.err6h[row] = e6h;
                if (e6n > 0.0f) atomicMax(&e6max_bits, __float_as_uint(e6n));
                if (e6h > 0.0f) atomicMax(&e6hmax_bits, __float_as_uint(e6h));
            }
            __syncthreads();
            const int u6t = tb6 / 16;   // a tile in 16-byte units; VFM_PREPARE_MX6_HALF stores the prefix the half-width pass stages
            const int u6p = o.mx6_half ? (MX6_SCALE_PLANE + (d >> 7) * MX6_KSTEP_BYTES) / 16 : u6t;
            uint4* dst = o.tiles6 + (int64_t)grp * (4 * u6t);
            const uint4* src = reinterpret_cast<const uint4*>(img6);
#ifdef VFM_PABL_NOST6
            for (int uu = threadIdx.x; uu < 4 * u6p && d < 0; uu += NT) {
#else
            for (int uu = threadIdx.x; uu < 4 * u6p; uu += NT) {
#endif
                const int u = (uu / u6p) * u6t + uu % u6p;
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
            for (int u = threadIdx.x; u < u16n; u += NT) {
                const uint4 tq = src[u];
                unsigned* po = reinterpret_cast<unsigned*>(dst + u);
                __builtin_nontemporal_store(tq.x, po);
                __builtin_nontemporal_store(tq.y, po + 1);
                __builtin_nontemporal_store(tq.z, po + 2);
                __builtin_nontemporal_store(tq.w, po + 3);
            }
        }
        if constexpr (MX6 && WAVES != 8) {
            if (gnext < groups) {
#pragma unroll
                for (int j = 0; j < RPW; ++j) load_row(gnext, j);
            }
        }
        if (threadIdx.x == 0) {
            o.gstep[grp] = qstep;
            o.gerr[grp] = __uint_as_float(emax_bits);
            o.grest[grp] = __uint_as_float(rmax_bits);
            if constexpr (MX6) {
                o.gerr6[grp] = __uint_as_float(e6max_bits);
                o.gerr6h[grp] = __uint_as_float(e6hmax_bits);
                o.gstep6[grp] = MX6_FIX_STEP;
                e6max_bits = 0u;
                e6hmax_bits = 0u;
            } else if (o.gerr6) {
                o.gerr6[grp] = o.gerr6h[grp] = __builtin_inff();
                o.gstep6[grp] = 0.0f;
            }
            amax_bits = 0u;   // for the next group (read again only behind the next two barriers)
            emax_bits = 0u;
            rmax_bits = 0u;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// The same images (int8 + fp6, d = 256 / 384) from a workgroup that fits BESIDE a workgroup of the coarse pass (round 4).
// prep_chunk_kernel keeps a group's 128 rows in registers between the pass that needs all of them (1 / |row|, the group's largest
// magnitude = its quantisation step) and the pass that quantises them: 8 waves x 200 registers and 150 KB of LDS -- a compute unit
// holds it or a workgroup of the fp6 coarse kernel (8 waves x 166 registers, 90 KB), never both, so in the pipeline the preparation of
// registration i + 1 and the coarse pass of registration i take turns on the compute units: a cycle is the SUM of the two (0.40 +
// 0.18 ms; profiles/r04_bench_kernel_stats.csv), with the matrix pipes idle for the length of the one and HBM idle for the other.
// Here the rows are read TWICE -- the second time mostly from the L2 / the memory-side cache: a workgroup re-reads its own 196 KB
// right behind the first pass -- in batches of 8 rows per wave, so nothing lives in registers across passes: 4 waves x <= 128
// registers, ~45 KB of LDS.  The dispatcher places such a workgroup on a compute unit that is running a coarse workgroup (332 of
// 512 registers per SIMD lane, 90 of 160 KB), its loads and VALU work run under the other's MFMAs.
//   wave w = rows 32 w .. of the group = one fragment tile of every image; batch b = rows 8 b .. of the tile.
//   pass 1: per batch the rows' sums of squares (oracle order: lane l owns chunks l, l + 64), 1 / |row| (scatter8), the largest
//           normalised magnitude; the batch's registers then take the next batch.
//   pass 2: per batch quantise (as prep_chunk_kernel: same arithmetic, same order -- the images and every E are bit-identical to
//           its), transpose through a 3 KB per-wave slice of the LDS and store 128-byte runs (8 rows of a 16-byte unit column);
//           the fp16 values of the columns the fp6 image covers go through the LDS the same way: lane = (row, 32-column block).
// ---------------------------------------------------------------------------------------------
template <int D, bool HALF>
__global__ __launch_bounds__(256, 3) void prep_stream_kernel(const float* __restrict__ x1, int64_t rows1, PrepOut o1, int groups1,
                                                         const float* __restrict__ x2, int64_t rows2, PrepOut o2, int groups) {
    constexpr int NC = D > 256 ? 2 : 1;          // float4 chunks per lane and row
    constexpr int NCH = D >> 2;                  // float4 chunks per row
    constexpr int NU = D >> 4;                   // 16-byte int8 units per row
    constexpr int NBLK = D >> 5;                 // 32-column blocks per row
    constexpr int RS8 = D + 16;                  // LDS bytes per int8 row of a batch (padded: the 8 rows of a unit column in 8 bank groups)
    constexpr int nconv = HALF ? NBLK >> 1 : NBLK;   // blocks the fp6 image covers (HALF: VFM_PREPARE_MX6_HALF, o.mx6_half)
    constexpr int nitems = 8 * nconv;            // (row, block) items of a batch
    __shared__ __attribute__((aligned(16))) unsigned char l_i8[4][8 * RS8];
    __shared__ __attribute__((aligned(16))) unsigned char l_h16[4][4 * nitems * 16];
    __shared__ float l_inv[I8_GROUP];
    __shared__ float l_e6[NBLK][I8_GROUP];
    __shared__ unsigned char l_sc[I8_GROUP][16];
    __shared__ unsigned amax_bits, emax_bits, rmax_bits, e6max_bits, e6hmax_bits;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int gidx = blockIdx.x;
    const bool second = gidx >= groups1;
    const int grp = second ? gidx - groups1 : gidx;
    const float* x = second ? x2 : x1;
    const int64_t rows = second ? rows2 : rows1;
    const PrepOut& o = second ? o2 : o1;
    if (threadIdx.x == 0) {
        amax_bits = 0u;
        emax_bits = 0u;
        rmax_bits = 0u;
        e6max_bits = 0u;
        e6hmax_bits = 0u;
    }
    float4 v[8][NC];
    const int64_t row0 = (int64_t)grp * I8_GROUP + wave * 32;   // first row of the wave's tile
    auto load_row = [&](int b, int j) __attribute__((always_inline)) {
        const int64_t r = row0 + 8 * b + j;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && c < NCH) {
                const float* pc = x + r * (int64_t)D + 4 * c;
                t.x = __builtin_nontemporal_load(pc);
                t.y = __builtin_nontemporal_load(pc + 1);
                t.z = __builtin_nontemporal_load(pc + 2);
                t.w = __builtin_nontemporal_load(pc + 3);
            }
            v[j][i] = t;
        }
    };
    // (first pass: plain loads -- the lines are wanted again in a few microseconds)
    auto load_row_keep = [&](int b, int j) __attribute__((always_inline)) {
        const int64_t r = row0 + 8 * b + j;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && c < NCH) t = *reinterpret_cast<const float4*>(x + r * (int64_t)D + 4 * c);
            v[j][i] = t;
        }
    };
    auto scatter8 = [&](const float* p) __attribute__((always_inline)) {   // (prep_chunk_kernel's: lane l ends with row l >> 3)
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
    // ---- pass 1
#pragma unroll
    for (int j = 0; j < 8; ++j) load_row_keep(0, j);
    float lmax = 0.0f;
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
        float part[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float p = 0.0f;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (lane + 64 * i < NCH) {
                    float t;
                    t = v[j][i].x * v[j][i].x; p = p + t;
                    t = v[j][i].y * v[j][i].y; p = p + t;
                    t = v[j][i].z * v[j][i].z; p = p + t;
                    t = v[j][i].w * v[j][i].w; p = p + t;
                }
            }
            part[j] = p;
        }
        const float my_inv = inv_norm_from_sumsq(scatter8(part));   // of row lane >> 3 of the batch
        if ((lane & 7) == 0) {
            l_inv[wave * 32 + 8 * b + (lane >> 3)] = my_inv;
            o.inv[row0 + 8 * b + (lane >> 3)] = my_inv;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float iv = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_inv), 8 * j));
#pragma unroll
            for (int i = 0; i < NC; ++i)
                lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(v[j][i].x * iv), fabsf(v[j][i].y * iv)), fmaxf(fabsf(v[j][i].z * iv), fabsf(v[j][i].w * iv))));
            if (b < 3) load_row_keep(b + 1, j);
            else load_row(0, j);   // the second pass's first batch
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    __syncthreads();   // the maxima are initialised
    if (lane == 0 && lmax > 0.0f) atomicMax(&amax_bits, __float_as_uint(lmax));
    __syncthreads();
    const float amax = __uint_as_float(amax_bits);
    const bool usable = amax > 0.0f && amax < 3.0e38f;
    const float qstep = usable ? amax / 127.0f : 1.0f;
    const float inv_qstep = usable ? 127.0f / amax : 0.0f;
    // ---- pass 2
    unsigned char* my8 = l_i8[wave];
    unsigned char* my16 = l_h16[wave];
    const int tb6 = mx6_tile_bytes(D >> 6);
    unsigned char* tile6 = reinterpret_cast<unsigned char*>(o.tiles6) + ((size_t)grp * 4 + wave) * (size_t)tb6;
    uint4* tile8 = o.tiles8 + ((size_t)grp * 4 + wave) * (size_t)(NU * 32);
    uint4* tile8h = o.tiles8h + ((size_t)grp * 4 + wave) * (size_t)((NU >> 1) * 32);
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
        float part[8], rpart[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float iv = l_inv[wave * 32 + 8 * b + j];
            float e2 = 0.0f, r2 = 0.0f;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int c = lane + 64 * i;
                if (c < NCH) {
                    // normalised values exactly as faiss leaves them in fp32
                    const float nv[4] = {v[j][i].x * iv, v[j][i].y * iv, v[j][i].z * iv, v[j][i].w * iv};
                    if (8 * c >= D) r2 = r2 + (nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2] + nv[3] * nv[3]);  // columns >= d / 2
                    int qi[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float qf = rintf(nv[e] * inv_qstep);
                        qf = fminf(fmaxf(qf, -127.0f), 127.0f);
                        const float res = __builtin_fmaf(-qstep, qf, nv[e]);
                        e2 = __builtin_fmaf(res, res, e2);
                        qi[e] = (int)qf;
                    }
                    const unsigned packed = __builtin_amdgcn_perm((unsigned)qi[1], (unsigned)qi[0], 0x0c0c0400u) |
                                            __builtin_amdgcn_perm((unsigned)qi[3], (unsigned)qi[2], 0x04000c0cu);
                    *reinterpret_cast<unsigned*>(my8 + j * RS8 + 4 * c) = packed;
                    if (c < 8 * nconv) {   // quarter k of (row, block): 8 halves = chunks 2 k, 2 k + 1 of the block
                        // (the fp32 product is rounded to fp16: hidden from the compiler, which otherwise turns product + conversion into
                        // one v_fma_mixlo_f16 -- a single rounding -- in SOME instantiations of this kernel: 1 % of the rows then carry an
                        // fp16 value one ulp off the definition, and E differs between the forms of the preparation)
                        float nh[4] = {nv[0], nv[1], nv[2], nv[3]};
                        asm volatile("" : "+v"(nh[0]), "+v"(nh[1]), "+v"(nh[2]), "+v"(nh[3]));
                        half4 h;
                        h[0] = (_Float16)nh[0];
                        h[1] = (_Float16)nh[1];
                        h[2] = (_Float16)nh[2];
                        h[3] = (_Float16)nh[3];
                        const int blk = c >> 3, k = (c & 7) >> 1, sub = c & 1;
                        *reinterpret_cast<half4*>(my16 + ((size_t)(k * nitems + j * nconv + blk) * 16 + sub * 8)) = h;
                    }
                }
            }
            part[j] = e2;
            rpart[j] = r2;
            if (b < 3) load_row(b + 1, j);
        }
        {
            float en = sqrtf(scatter8(part)) * 1.000244140625f + 1.0e-30f;
            if (!(en == en)) en = __builtin_inff();
            float rn = sqrtf(scatter8(rpart)) * 1.000244140625f + 1.0e-30f;
            if (!(rn == rn)) rn = __builtin_inff();
            const int64_t r = row0 + 8 * b + (lane >> 3);
            if (r >= rows) en = rn = 0.0f;
            if ((lane & 7) == 0) {
                o.err[r] = en;
                o.rest[r] = rn;
                if (en > 0.0f) atomicMax(&emax_bits, __float_as_uint(en));
                if (rn > 0.0f) atomicMax(&rmax_bits, __float_as_uint(rn));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // int8: unit column u of the batch's 8 rows = 128 consecutive bytes of the tile
#pragma unroll
        for (int rd = 0; rd < (NU * 8 + 63) / 64; ++rd) {
            const int e = rd * 64 + lane, u = e >> 3, j = e & 7;
            if (e < NU * 8) {
                const uint4 tq = *reinterpret_cast<const uint4*>(my8 + j * RS8 + 16 * u);
                unsigned* po = reinterpret_cast<unsigned*>(tile8 + u * 32 + 8 * b + j);
                __builtin_nontemporal_store(tq.x, po);
                __builtin_nontemporal_store(tq.y, po + 1);
                __builtin_nontemporal_store(tq.z, po + 2);
                __builtin_nontemporal_store(tq.w, po + 3);
                if (!HALF && u < (NU >> 1)) {   // the first d / 2 columns again, as tiles of their own
                    unsigned* ph = reinterpret_cast<unsigned*>(tile8h + u * 32 + 8 * b + j);
                    __builtin_nontemporal_store(tq.x, ph);
                    __builtin_nontemporal_store(tq.y, ph + 1);
                    __builtin_nontemporal_store(tq.z, ph + 2);
                    __builtin_nontemporal_store(tq.w, ph + 3);
                }
            }
        }
        if (o.rows8) {   // row-major copy (scan-sized operands): the batch's rows as they lie in the slice
#pragma unroll
            for (int rd = 0; rd < (NU * 8 + 63) / 64; ++rd) {
                const int e = rd * 64 + lane, j = e / NU, u = e - j * NU;
                if (e < NU * 8)
                    *reinterpret_cast<uint4*>(o.rows8 + (size_t)(row0 + 8 * b + j) * D + 16 * u) = *reinterpret_cast<const uint4*>(my8 + j * RS8 + 16 * u);
            }
        }
        // fp6: lane = (row j, block) of the batch -- prep_chunk_kernel's conversion
#pragma unroll
        for (int item0 = 0; item0 < nitems; item0 += 64) {
            const int item = item0 + lane;
            if (item >= nitems) break;
            const int j = item / nconv, blk = item - j * nconv, p = 8 * b + j;
            union {
                uint4 u[4];
                halfx32 h;
                unsigned w[16];
            } vv;
#pragma unroll
            for (int k = 0; k < 4; ++k) vv.u[k] = *reinterpret_cast<const uint4*>(my16 + (size_t)(k * nitems + item) * 16);
            ushortx2 m2 = {0, 0};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned a2 = vv.w[i] & 0x7fff7fffu;
                m2 = __builtin_elementwise_max(m2, *reinterpret_cast<const ushortx2*>(&a2));
            }
            const unsigned am = max((unsigned)m2[0], (unsigned)m2[1]);
            int ex = (int)(am >> 10) - 15 - ((am & 0x3ffu) <= 0x3c0u ? 2 : 1);
            ex = ex > 0 ? 0 : ex;
            const float sc6 = __uint_as_float((unsigned)(ex + 127) << 23);
            const intx6 codes = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vv.h, sc6);
            const halfx32 back = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(codes, 1.0f);
            float e6 = 0.0f;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float res = __builtin_fmaf(-(float)back[i], sc6, (float)vv.h[i]);
                e6 = __builtin_fmaf(res, res, e6);
            }
            const int s6 = blk >> 1, l6 = (blk & 1) * 32 + p;
            unsigned* pa = reinterpret_cast<unsigned*>(tile6 + mx6_code_a(s6, l6));
            __builtin_nontemporal_store((unsigned)codes[0], pa);
            __builtin_nontemporal_store((unsigned)codes[1], pa + 1);
            __builtin_nontemporal_store((unsigned)codes[2], pa + 2);
            __builtin_nontemporal_store((unsigned)codes[3], pa + 3);
            unsigned* pb = reinterpret_cast<unsigned*>(tile6 + mx6_code_b(s6, l6));
            __builtin_nontemporal_store((unsigned)codes[4], pb);
            __builtin_nontemporal_store((unsigned)codes[5], pb + 1);
            l_sc[wave * 32 + p][blk] = (unsigned char)(ex + 127);
            l_e6[blk][wave * 32 + p] = e6;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // (the batch's slices of the LDS are read: the next batch may overwrite them)
    }
    __syncthreads();
    {   // the d / 64 scales of MFMA lane (hh, p) of tile t: its 8 bytes of the scale plane
        const int r = threadIdx.x & (I8_GROUP - 1), hh = threadIdx.x >> 7;
        unsigned lo = 0u, hi = 0u;
        for (int s6 = 0; s6 < (nconv >> 1); ++s6) {
            const unsigned bsc = l_sc[r][2 * s6 + hh];
            if (s6 < 4) lo |= bsc << (8 * s6);
            else hi |= bsc << (8 * (s6 - 4));
        }
        unsigned char* t6 = reinterpret_cast<unsigned char*>(o.tiles6) + ((size_t)grp * 4 + (r >> 5)) * (size_t)tb6;
        *reinterpret_cast<uint2*>(t6 + mx6_scale_at(D >> 6, 0, hh * 32 + (r & 31))) = make_uint2(lo, hi);
    }
    if (threadIdx.x < I8_GROUP) {   // E of the fp6 image per row: blocks in order; rounded up like the int8 one
        const int r = threadIdx.x;
        float acc = 0.0f, acch = 0.0f;
        for (int blk = 0; blk < nconv; ++blk) {
            acc = acc + l_e6[blk][r];
            if (blk + 1 == (NBLK >> 1)) acch = acc;
        }
        float e6n = sqrtf(acc) * 1.000244140625f + (MX6_F16_ROUNDING + MX6_SLACK);
        float e6h = sqrtf(acch) * 1.000244140625f + (MX6_F16_ROUNDING + MX6_SLACK);
        if (!(e6n == e6n) || HALF) e6n = __builtin_inff();
        if (!(e6h == e6h)) e6h = __builtin_inff();
        const int64_t row = (int64_t)grp * I8_GROUP + r;
        if (row >= rows) e6n = e6h = 0.0f;
        o.err6[row] = e6n;
        o.err6h[row] = e6h;
        if (e6n > 0.0f) atomicMax(&e6max_bits, __float_as_uint(e6n));
        if (e6h > 0.0f) atomicMax(&e6hmax_bits, __float_as_uint(e6h));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        o.gstep[grp] = qstep;
        o.gerr[grp] = __uint_as_float(emax_bits);
        o.grest[grp] = __uint_as_float(rmax_bits);
        o.gerr6[grp] = __uint_as_float(e6max_bits);
        o.gerr6h[grp] = __uint_as_float(e6hmax_bits);
        o.gstep6[grp] = MX6_FIX_STEP;
    }
}

// ---------------------------------------------------------------------------------------------
// The same images from ONE read of the rows by a workgroup of prep_stream_kernel's size (round 6; VERDICT r5 item 3c: the stream form
// moves 782 MB -- every row twice through the fabric, the second time from the L2 / the memory-side cache -- for 462 MB that must
// move once, and in the pipeline a cycle is coarse kernel + preparation).  What has to survive from the pass that needs all 128 rows
// of a group (1 / |row|, the group's largest magnitude = its quantisation step) to the pass that quantises is the NORMALISED row, and
// the fp6 image is defined on its fp16 rounding anyway: pass 1 leaves the wave's 32 rows as packed halves in REGISTERS -- 3 registers
// per row at d = 384 (a lane's first chunk: 2; the second chunks of a pair of rows share 2: lanes 0 - 31 the even row's, lanes 32 - 63
// the odd row's, moved there by v_permlane32_swap), 96 in all -- beside a batch of 8 rows in flight; nothing is read twice and no
// group lives in the LDS (prep_chunk_kernel: 8 waves x 200 registers + 150 KB, one workgroup per compute unit, 0.185 ms alone).
// Definition of THIS form's int8 image (it is not byte-identical to the other two forms'): with h = fp16(v), v the fp32-normalised
// element, q = clamp(rint(h * 127 / amax)), amax the group's largest |v| in fp32 as before; E = |h - s q|_2 + |v - h|_2, both terms
// MEASURED (the second in pass 1, where v and h are both at hand; ~1.4e-4 beside ~1e-2), each rounded up.  The fp6 image, err6 / err6h,
// 1 / |row|, rest / grest and every group datum are the other forms', bit for bit (same arithmetic on the same values).
// ---------------------------------------------------------------------------------------------
template <int D, bool HALF>
__global__ __launch_bounds__(256, 2) void prep_once_kernel(const float* __restrict__ x1, int64_t rows1, PrepOut o1, int groups1,
                                                           const float* __restrict__ x2, int64_t rows2, PrepOut o2, int groups) {
    constexpr bool PAIR = D > 256;               // d = 384: a row is one chunk per lane + half a chunk -- the second chunks of TWO rows share a register
    constexpr int NU = D >> 4;                   // 16-byte int8 units per row
    constexpr int NBLK = D >> 5;                 // 32-column blocks per row
    constexpr int RS8 = D + 16;                  // LDS bytes per int8 row of a batch (padded: the 8 rows of a unit column in 8 bank groups)
    constexpr int nconv = HALF ? NBLK >> 1 : NBLK;   // blocks the fp6 image covers (HALF: VFM_PREPARE_MX6_HALF, o.mx6_half)
    constexpr int nitems = 8 * nconv;            // (row, block) items of a batch
    static_assert(D == 256 || D == 384, "a lane owns one chunk (d = 256) or one and a half (d = 384) of a row");
    __shared__ __attribute__((aligned(16))) unsigned char l_i8[4][8 * RS8];
    __shared__ __attribute__((aligned(16))) unsigned char l_h16[4][4 * nitems * 16];
    __shared__ float l_e6[NBLK][I8_GROUP];
    __shared__ unsigned char l_sc[I8_GROUP][16];
    __shared__ unsigned amax_bits, emax_bits, rmax_bits, e6max_bits, e6hmax_bits;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const bool lo = lane < 32;
    const int gidx = blockIdx.x;
    const bool second = gidx >= groups1;
    const int grp = second ? gidx - groups1 : gidx;
    const float* x = second ? x2 : x1;
    const int64_t rows = second ? rows2 : rows1;
    const PrepOut& o = second ? o2 : o1;
    if (threadIdx.x == 0) {
        amax_bits = 0u;
        emax_bits = 0u;
        rmax_bits = 0u;
        e6max_bits = 0u;
        e6hmax_bits = 0u;
    }
    // a batch of 8 rows in flight: v0[j] = the lane's first chunk (columns 4 lane ..) of row j; vp[jp] (d = 384) = the second chunk --
    // columns 256 + 4 (lane & 31) .. -- of row 2 jp (lanes 0 - 31) and of row 2 jp + 1 (lanes 32 - 63): every lane of every load, every
    // multiply and every conversion below works on a real element (with one register per row the upper half of the wave carried zeros
    // through a quarter of the kernel's VALU work)
    float4 v0[8];
    float4 vp[PAIR ? 4 : 1];
    const int64_t row0 = (int64_t)grp * I8_GROUP + wave * 32;   // first row of the wave's tile
    auto load4 = [&](int64_t r, int c) __attribute__((always_inline)) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef VFM_POABL_NOLOAD   // (timing experiments, tools/build_ablate_prep.sh: results are garbage)
        if (r < rows) t = make_float4(1.0f + (float)c, 2.0f - (float)(r & 7), 0.5f * (float)lane, 3.0f);
#else
        if (r < rows) {
            const float* pc = x + r * (int64_t)D + 4 * c;
            t.x = __builtin_nontemporal_load(pc);
            t.y = __builtin_nontemporal_load(pc + 1);
            t.z = __builtin_nontemporal_load(pc + 2);
            t.w = __builtin_nontemporal_load(pc + 3);
        }
#endif
        return t;
    };
    auto load_row = [&](int b, int j) __attribute__((always_inline)) { v0[j] = load4(row0 + 8 * b + j, lane); };
    auto load_pair = [&](int b, int jp) __attribute__((always_inline)) { vp[jp] = load4(row0 + 8 * b + 2 * jp + (lane >> 5), 64 + (lane & 31)); };
    auto scatter8 = [&](const float* p) __attribute__((always_inline)) {   // (prep_chunk_kernel's: lane l ends with row l >> 3)
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
    // v_permlane32_swap(a, b): result 0 = {a lanes 0-31 | b lanes 0-31}, result 1 = {a lanes 32-63 | b lanes 32-63} (rows of 32 lanes)
    auto lo_lo = [&](float a, float b) __attribute__((always_inline)) {   // lanes 0-31: a's lower half; lanes 32-63: b's LOWER half
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        return __uint_as_float((unsigned)sw[0]);
    };
    auto hi_to_lo = [&](float a) __attribute__((always_inline)) {         // lanes 0-31: a's UPPER half (lanes 32-63: a's own)
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
        return __uint_as_float((unsigned)sw[1]);
    };
    // the wave's 32 normalised rows as packed halves: hs0[row] = the lane's first chunk; hs1[pair] (d = 384) = the pair's second chunks
    uint2 hs0[32];
    uint2 hs1[PAIR ? 16 : 1];
    auto pack4 = [&](const float (&nv)[4]) __attribute__((always_inline)) {
        // (the fp32 product is rounded to fp16 as a value of its own -- see prep_stream_kernel: hidden from the compiler, which otherwise
        // fuses product and conversion into one v_fma_mixlo_f16, a single rounding)
        float nh[4] = {nv[0], nv[1], nv[2], nv[3]};
        asm volatile("" : "+v"(nh[0]), "+v"(nh[1]), "+v"(nh[2]), "+v"(nh[3]));
        half4 h;
        h[0] = (_Float16)nh[0];
        h[1] = (_Float16)nh[1];
        h[2] = (_Float16)nh[2];
        h[3] = (_Float16)nh[3];
        uint2 u;
        __builtin_memcpy(&u, &h, 8);
        // (as two PACKED registers from here on: left to itself the compiler keeps the four halves in four registers until their last use
        // -- the 96-register copy of the tile became 192 and 262 registers were spilled)
        asm volatile("" : "+v"(u.x), "+v"(u.y));
        return u;
    };
    auto unpack4 = [&](uint2 u, float (&nv)[4]) __attribute__((always_inline)) {
        half4 h;
        __builtin_memcpy(&h, &u, 8);
        nv[0] = (float)h[0];
        nv[1] = (float)h[1];
        nv[2] = (float)h[2];
        nv[3] = (float)h[3];
    };
    // sum of a chunk's squares onto p in the oracle's order: ((((p + x^2) + y^2) + z^2) + w^2)
    auto fold4 = [&](float p, const float4& t) __attribute__((always_inline)) {
        float q;
        q = t.x * t.x; p = p + q;
        q = t.y * t.y; p = p + q;
        q = t.z * t.z; p = p + q;
        q = t.w * t.w; p = p + q;
        return p;
    };
    // ---- pass 1: 1 / |row| (oracle order), the group's largest normalised magnitude, |second half|, the fp16 copy
#pragma unroll
    for (int j = 0; j < 8; ++j) load_row(0, j);
    if constexpr (PAIR) {
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) load_pair(0, jp);
    }
    __syncthreads();   // the maxima are initialised (pass 1 already adds to rmax_bits)
    float lmax = 0.0f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        __builtin_amdgcn_sched_barrier(0);   // (the four unrolled batches stay apart)
        float part[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) part[j] = fold4(0.0f, v0[j]);
        if constexpr (PAIR) {
            // lane l < 32 of a row goes on with its chunk 64 + l: the even row's continues in place, the odd row's in lane l + 32, where its
            // second chunk was loaded -- one fold for both rows -- and comes back to lane l for the butterfly (whose tree is the oracle's)
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                const float s = fold4(lo_lo(part[2 * jp], part[2 * jp + 1]), vp[jp]);
                const float sb = hi_to_lo(s);
                part[2 * jp] = lo ? s : part[2 * jp];
                part[2 * jp + 1] = lo ? sb : part[2 * jp + 1];
            }
        }
        const float my_inv = inv_norm_from_sumsq(scatter8(part));   // of row lane >> 3 of the batch
        if ((lane & 7) == 0) o.inv[row0 + 8 * b + (lane >> 3)] = my_inv;
        float rpart[8], ivs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float iv = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(my_inv), 8 * j));
            ivs[j] = iv;
            // normalised values exactly as faiss leaves them in fp32
            const float nv[4] = {v0[j].x * iv, v0[j].y * iv, v0[j].z * iv, v0[j].w * iv};
            lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(nv[0]), fabsf(nv[1])), fmaxf(fabsf(nv[2]), fabsf(nv[3]))));
            const float ss = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2] + nv[3] * nv[3];
            float r2 = 0.0f;
            r2 = r2 + ((8 * lane >= D) ? ss : 0.0f);   // columns >= d / 2 (the other forms' sum: x + 0 = x)
            hs0[8 * b + j] = pack4(nv);
            // (the row's sums are wanted HERE: left to the compiler they were formed at the end of the batch and the normalised values of
            // its rows were spilled to wait for it)
            asm volatile("" : "+v"(r2), "+v"(lmax));
            rpart[j] = r2;
            if (b < 3) load_row(b + 1, j);
        }
        if constexpr (PAIR) {
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                const float iv = lo ? ivs[2 * jp] : ivs[2 * jp + 1];
                const float nv[4] = {vp[jp].x * iv, vp[jp].y * iv, vp[jp].z * iv, vp[jp].w * iv};
                lmax = fmaxf(lmax, fmaxf(fmaxf(fabsf(nv[0]), fabsf(nv[1])), fmaxf(fabsf(nv[2]), fabsf(nv[3]))));
                float ss = nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2] + nv[3] * nv[3];   // all of these columns are >= d / 2
                hs1[4 * b + jp] = pack4(nv);
                asm volatile("" : "+v"(ss), "+v"(lmax));
                const float ssb = hi_to_lo(ss);                       // the odd row's, back in the lanes the oracle's tree expects
                rpart[2 * jp] = rpart[2 * jp] + (lo ? ss : 0.0f);
                rpart[2 * jp + 1] = rpart[2 * jp + 1] + (lo ? ssb : 0.0f);
                if (b < 3) load_pair(b + 1, jp);
            }
        }
        {
            float rn = sqrtf(scatter8(rpart)) * 1.000244140625f + 1.0e-30f;
            if (!(rn == rn)) rn = __builtin_inff();
            const int64_t r = row0 + 8 * b + (lane >> 3);
            if (r >= rows) rn = 0.0f;
            if ((lane & 7) == 0) {
                o.rest[r] = rn;
                if (rn > 0.0f) atomicMax(&rmax_bits, __float_as_uint(rn));
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    __syncthreads();
    if (lane == 0 && lmax > 0.0f) atomicMax(&amax_bits, __float_as_uint(lmax));
    __syncthreads();
    const float amax = __uint_as_float(amax_bits);
    const bool usable = amax > 0.0f && amax < 3.0e38f;
    const float qstep = usable ? amax / 127.0f : 1.0f;
    const float inv_qstep = usable ? 127.0f / amax : 0.0f;
    // ---- pass 2: from the registers
    unsigned char* my8 = l_i8[wave];
    unsigned char* my16 = l_h16[wave];
    const int tb6 = mx6_tile_bytes(D >> 6);
    unsigned char* tile6 = reinterpret_cast<unsigned char*>(o.tiles6) + ((size_t)grp * 4 + wave) * (size_t)tb6;
    uint4* tile8 = o.tiles8 + ((size_t)grp * 4 + wave) * (size_t)(NU * 32);
    uint4* tile8h = o.tiles8h + ((size_t)grp * 4 + wave) * (size_t)((NU >> 1) * 32);
    // one chunk of a row: quantise (codes to the batch's int8 slice), residual against the fp16 value, the halves to the fp6 staging
    auto quant_chunk = [&](uint2 hh, int jr, int c) __attribute__((always_inline)) {
        float nv[4];
        unpack4(hh, nv);
        int qi[4];
        float e2 = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (no clamp: |h| <= fp16(amax) <= amax (1 + 2^-11) and inv_qstep <= (127 / amax)(1 + 2^-24), so |h inv_qstep| < 127.07 and the
            // rounded value is within [-127, 127] by itself -- two of this loop's eight VALU operations per element, in a kernel the
            // ablations put on the VALU (profiles/r06_ablations_prep_once.txt).  A NaN element converts to code 0; its row's E is infinite.)
            const float qf = rintf(nv[e] * inv_qstep);
            const float res = __builtin_fmaf(-qstep, qf, nv[e]);
            e2 = __builtin_fmaf(res, res, e2);
            qi[e] = (int)qf;
        }
        const unsigned packed = __builtin_amdgcn_perm((unsigned)qi[1], (unsigned)qi[0], 0x0c0c0400u) |
                                __builtin_amdgcn_perm((unsigned)qi[3], (unsigned)qi[2], 0x04000c0cu);
        *reinterpret_cast<unsigned*>(my8 + jr * RS8 + 4 * c) = packed;
        if (c < 8 * nconv) {   // quarter k of (row, block): 8 halves = chunks 2 k, 2 k + 1 of the block
            const int blk = c >> 3, k = (c & 7) >> 1, sub = c & 1;
            *reinterpret_cast<uint2*>(my16 + ((size_t)(k * nitems + jr * nconv + blk) * 16 + sub * 8)) = hh;
        }
        return e2;
    };
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        __builtin_amdgcn_sched_barrier(0);
        float part[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) part[j] = quant_chunk(hs0[8 * b + j], j, lane);
        if constexpr (PAIR) {
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                const float e2 = quant_chunk(hs1[4 * b + jp], 2 * jp + (lane >> 5), 64 + (lane & 31));
                // (the residual's sum has no prescribed order: the odd row's share stays in the upper lanes)
                part[2 * jp] = part[2 * jp] + (lo ? e2 : 0.0f);
                part[2 * jp + 1] = part[2 * jp + 1] + (lo ? 0.0f : e2);
            }
        }
        {
            // E = |h - s q|_2 (measured, rounded up) + |v - h|_2 (bounded: PREP_F16_ROUNDING)
            float en = (sqrtf(scatter8(part)) * 1.000244140625f + PREP_F16_ROUNDING) * 1.0000002384185791f + 1.0e-30f;
            if (!(en == en)) en = __builtin_inff();
            const int64_t r = row0 + 8 * b + (lane >> 3);
            if (r >= rows) en = 0.0f;
            if ((lane & 7) == 0) {
                o.err[r] = en;
                if (en > 0.0f) atomicMax(&emax_bits, __float_as_uint(en));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // int8: unit column u of the batch's 8 rows = 128 consecutive bytes of the tile
#ifndef VFM_POABL_NOST8
#pragma unroll
        for (int rd = 0; rd < (NU * 8 + 63) / 64; ++rd) {
            const int e = rd * 64 + lane, u = e >> 3, j = e & 7;
            if (e < NU * 8) {
                const uint4 tq = *reinterpret_cast<const uint4*>(my8 + j * RS8 + 16 * u);
                unsigned* po = reinterpret_cast<unsigned*>(tile8 + u * 32 + 8 * b + j);
                __builtin_nontemporal_store(tq.x, po);
                __builtin_nontemporal_store(tq.y, po + 1);
                __builtin_nontemporal_store(tq.z, po + 2);
                __builtin_nontemporal_store(tq.w, po + 3);
                if (!HALF && u < (NU >> 1)) {   // the first d / 2 columns again, as tiles of their own
                    unsigned* ph = reinterpret_cast<unsigned*>(tile8h + u * 32 + 8 * b + j);
                    __builtin_nontemporal_store(tq.x, ph);
                    __builtin_nontemporal_store(tq.y, ph + 1);
                    __builtin_nontemporal_store(tq.z, ph + 2);
                    __builtin_nontemporal_store(tq.w, ph + 3);
                }
            }
        }
#endif
        if (o.rows8) {   // row-major copy (scan-sized operands): the batch's rows as they lie in the slice
#pragma unroll
            for (int rd = 0; rd < (NU * 8 + 63) / 64; ++rd) {
                const int e = rd * 64 + lane, j = e / NU, u = e - j * NU;
                if (e < NU * 8)
                    *reinterpret_cast<uint4*>(o.rows8 + (size_t)(row0 + 8 * b + j) * D + 16 * u) = *reinterpret_cast<const uint4*>(my8 + j * RS8 + 16 * u);
            }
        }
        // fp6: lane = (row j, block) of the batch -- prep_chunk_kernel's conversion
#ifndef VFM_POABL_NOMX6
#pragma unroll
        for (int item0 = 0; item0 < nitems; item0 += 64) {
            const int item = item0 + lane;
            if (item >= nitems) break;
            const int j = item / nconv, blk = item - j * nconv, p = 8 * b + j;
            union {
                uint4 u[4];
                halfx32 h;
                unsigned w[16];
            } vv;
#pragma unroll
            for (int k = 0; k < 4; ++k) vv.u[k] = *reinterpret_cast<const uint4*>(my16 + (size_t)(k * nitems + item) * 16);
            ushortx2 m2 = {0, 0};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned a2 = vv.w[i] & 0x7fff7fffu;
                m2 = __builtin_elementwise_max(m2, *reinterpret_cast<const ushortx2*>(&a2));
            }
            const unsigned am = max((unsigned)m2[0], (unsigned)m2[1]);
            int ex = (int)(am >> 10) - 15 - ((am & 0x3ffu) <= 0x3c0u ? 2 : 1);
            ex = ex > 0 ? 0 : ex;
            const float sc6 = __uint_as_float((unsigned)(ex + 127) << 23);
            const intx6 codes = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vv.h, sc6);
            const halfx32 back = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(codes, 1.0f);
            float e6 = 0.0f;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float res = __builtin_fmaf(-(float)back[i], sc6, (float)vv.h[i]);
                e6 = __builtin_fmaf(res, res, e6);
            }
            const int s6 = blk >> 1, l6 = (blk & 1) * 32 + p;
            unsigned* pa = reinterpret_cast<unsigned*>(tile6 + mx6_code_a(s6, l6));
            __builtin_nontemporal_store((unsigned)codes[0], pa);
            __builtin_nontemporal_store((unsigned)codes[1], pa + 1);
            __builtin_nontemporal_store((unsigned)codes[2], pa + 2);
            __builtin_nontemporal_store((unsigned)codes[3], pa + 3);
            unsigned* pb = reinterpret_cast<unsigned*>(tile6 + mx6_code_b(s6, l6));
            __builtin_nontemporal_store((unsigned)codes[4], pb);
            __builtin_nontemporal_store((unsigned)codes[5], pb + 1);
            l_sc[wave * 32 + p][blk] = (unsigned char)(ex + 127);
            l_e6[blk][wave * 32 + p] = e6;
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // (the batch's slices of the LDS are read: the next batch may overwrite them)
    }
    __syncthreads();
    {   // the d / 64 scales of MFMA lane (hh, p) of tile t: its 8 bytes of the scale plane
        const int r = threadIdx.x & (I8_GROUP - 1), hh = threadIdx.x >> 7;
        unsigned lo4 = 0u, hi4 = 0u;
        for (int s6 = 0; s6 < (nconv >> 1); ++s6) {
            const unsigned bsc = l_sc[r][2 * s6 + hh];
            if (s6 < 4) lo4 |= bsc << (8 * s6);
            else hi4 |= bsc << (8 * (s6 - 4));
        }
        unsigned char* t6 = reinterpret_cast<unsigned char*>(o.tiles6) + ((size_t)grp * 4 + (r >> 5)) * (size_t)tb6;
        *reinterpret_cast<uint2*>(t6 + mx6_scale_at(D >> 6, 0, hh * 32 + (r & 31))) = make_uint2(lo4, hi4);
    }
    if (threadIdx.x < I8_GROUP) {   // E of the fp6 image per row: blocks in order; rounded up like the int8 one
        const int r = threadIdx.x;
        float acc = 0.0f, acch = 0.0f;
        for (int blk = 0; blk < nconv; ++blk) {
            acc = acc + l_e6[blk][r];
            if (blk + 1 == (NBLK >> 1)) acch = acc;
        }
        float e6n = sqrtf(acc) * 1.000244140625f + (MX6_F16_ROUNDING + MX6_SLACK);
        float e6h = sqrtf(acch) * 1.000244140625f + (MX6_F16_ROUNDING + MX6_SLACK);
        if (!(e6n == e6n) || HALF) e6n = __builtin_inff();
        if (!(e6h == e6h)) e6h = __builtin_inff();
        const int64_t row = (int64_t)grp * I8_GROUP + r;
        if (row >= rows) e6n = e6h = 0.0f;
        o.err6[row] = e6n;
        o.err6h[row] = e6h;
        if (e6n > 0.0f) atomicMax(&e6max_bits, __float_as_uint(e6n));
        if (e6h > 0.0f) atomicMax(&e6hmax_bits, __float_as_uint(e6h));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        o.gstep[grp] = qstep;
        o.gerr[grp] = __uint_as_float(emax_bits);
        o.grest[grp] = __uint_as_float(rmax_bits);
        o.gerr6[grp] = __uint_as_float(e6max_bits);
        o.gerr6h[grp] = __uint_as_float(e6hmax_bits);
        o.gstep6[grp] = MX6_FIX_STEP;
    }
}

// ---------------------------------------------------------------------------------------------
// The fp6 image for the wider rows (d = 512, 768: the half-width pass in fp6 reads its first d / 128 k-steps), by kernels of their
// own behind prep_chunk_kernel (whose LDS cannot hold an fp16 copy of such a group): a thread takes one (row, 32-column block),
// reads its 32 floats, normalises them with the 1 / |row| prep_chunk_kernel left, rounds to fp16 and converts exactly as that
// kernel's second half does (same scale rule, v_cvt_scalef32_pk32_fp6_f16, residual measured against the fp16 values); the blocks
// of a row are NBLK consecutive lanes, so the row's sum is a fixed xor tree.  The block scales of k-steps 0 .. 7 go to the tile's
// first scale plane, those of k-steps 8 .. 11 to the second one behind the codes (mx6_scale_at; spare scale bytes are never
// read).  prep_mx6_group_kernel then takes the maximum E of every group.
// ---------------------------------------------------------------------------------------------
template <int NBLK>   // lanes per row: 16 (d = 512) or 32 (d = 768: 24 in use)
__global__ __launch_bounds__(256) void prep_mx6_rows_kernel(Rows x1, int64_t rows1, int64_t pad1, int d, PrepOut o1,
                                                            Rows x2, int64_t rows2, PrepOut o2, int64_t total) {
    constexpr int RPB = 256 / NBLK;   // rows per workgroup
    const int blk = threadIdx.x % NBLK;
    int64_t g = (int64_t)blockIdx.x * RPB + threadIdx.x / NBLK;   // row of the padded concatenation
    const bool beyond = g >= total;   // (total is a multiple of 256 rows: never with RPB = 8 / 16; kept for safety -- no early return before the barrier)
    if (beyond) g = total - 1;
    const bool second = g >= pad1;
    const int64_t r = second ? g - pad1 : g;
    const int64_t rows = second ? rows2 : rows1;
    const PrepOut& o = second ? o2 : o1;
    const int nblk = d >> 5, nconv = o.mx6_half ? nblk >> 1 : nblk;   // VFM_PREPARE_MX6_HALF: the first d / 2 columns only
    const bool active = blk < nconv && !beyond;
    // the workgroup's RPB rows come in coalesced (consecutive threads, consecutive float4) and go through the LDS to the thread
    // that owns their block: float4 j of block b at slot 8 b + (j ^ (b & 7)) of its row (the xor keeps the 128-byte-strided
    // reads of a half-wave off one bank group)
    extern __shared__ __attribute__((aligned(16))) unsigned char mx6_smem[];
    float4* stage = reinterpret_cast<float4*>(mx6_smem);   // [RPB][d / 4]
    const int nf4 = d >> 2;
    const int64_t g0 = (int64_t)blockIdx.x * RPB;
    for (int idx = threadIdx.x; idx < RPB * nf4; idx += 256) {
        const int rr = idx / nf4, c4 = idx % nf4;
        const int64_t gg = g0 + rr;
        const bool sec = gg >= pad1;
        const int64_t r2 = sec ? gg - pad1 : gg;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gg < total && r2 < (sec ? rows2 : rows1)) {
            t = (sec ? x2 : x1).ld4_nt(r2 * (int64_t)d + 4 * c4);
        }
        const int bb = c4 >> 3, jj = c4 & 7;
        stage[rr * nf4 + bb * 8 + (jj ^ (bb & 7))] = t;
    }
    __syncthreads();
    union {
        halfx32 h;
        unsigned w[16];
    } v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v.w[i] = 0u;
    if (active && r < rows) {
        const float iv = o.inv[r];
        const float4* mine = stage + (threadIdx.x / NBLK) * nf4 + blk * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 t = mine[i ^ (blk & 7)];
            v.h[4 * i + 0] = (_Float16)(t.x * iv);
            v.h[4 * i + 1] = (_Float16)(t.y * iv);
            v.h[4 * i + 2] = (_Float16)(t.z * iv);
            v.h[4 * i + 3] = (_Float16)(t.w * iv);
        }
    }
    ushortx2 m2 = {0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned a2 = v.w[i] & 0x7fff7fffu;
        m2 = __builtin_elementwise_max(m2, *reinterpret_cast<const ushortx2*>(&a2));
    }
    const unsigned am = max((unsigned)m2[0], (unsigned)m2[1]);
    int ex = (int)(am >> 10) - 15 - ((am & 0x3ffu) <= 0x3c0u ? 2 : 1);   // as in prep_chunk_kernel's conversion
    ex = ex > 0 ? 0 : ex;
    const float sc6 = __uint_as_float((unsigned)(ex + 127) << 23);
    const intx6 codes = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v.h, sc6);
    const halfx32 back = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(codes, 1.0f);
    float e6 = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float res = __builtin_fmaf(-(float)back[i], sc6, (float)v.h[i]);
        e6 = __builtin_fmaf(res, res, e6);
    }
    if (!active) e6 = 0.0f;
    float e6f = e6, e6hs = blk < (nblk >> 1) ? e6 : 0.0f;   // all converted columns / the first d / 2
#pragma unroll
    for (int off = NBLK / 2; off >= 1; off >>= 1) {
        e6f = e6f + __shfl_xor(e6f, off);
        e6hs = e6hs + __shfl_xor(e6hs, off);
    }
    if (blk == 0 && !beyond) {
        float e6n = sqrtf(e6f) * 1.000244140625f + (MX6_F16_ROUNDING + MX6_SLACK);
        float e6h = sqrtf(e6hs) * 1.000244140625f + (MX6_F16_ROUNDING + MX6_SLACK);
        if (!(e6n == e6n) || o.mx6_half) e6n = __builtin_inff();
        if (!(e6h == e6h)) e6h = __builtin_inff();
        if (r >= rows) e6n = e6h = 0.0f;
        o.err6[r] = e6n;
        o.err6h[r] = e6h;
    }
    if (active) {
        const int p = (int)(r & 31), s6 = blk >> 1, l6 = (blk & 1) * 32 + p, ks = d >> 6;
        unsigned char* tile = reinterpret_cast<unsigned char*>(o.tiles6) + (size_t)(r >> 5) * (size_t)mx6_tile_bytes(ks);
        *reinterpret_cast<uint4*>(tile + mx6_code_a(s6, l6)) = make_uint4((unsigned)codes[0], (unsigned)codes[1], (unsigned)codes[2], (unsigned)codes[3]);
        *reinterpret_cast<uint2*>(tile + mx6_code_b(s6, l6)) = make_uint2((unsigned)codes[4], (unsigned)codes[5]);
        tile[mx6_scale_at(ks, s6, l6)] = (unsigned char)(ex + 127);
    }
}

__global__ __launch_bounds__(256) void prep_mx6_group_kernel(PrepOut o1, int groups1, PrepOut o2, int groups) {
    const int gidx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gidx >= groups) return;
    const bool second = gidx >= groups1;
    const PrepOut& o = second ? o2 : o1;
    const int grp = second ? gidx - groups1 : gidx;
    const int lane = lane_id();
    float m = fmaxf(o.err6[(int64_t)grp * I8_GROUP + lane], o.err6[(int64_t)grp * I8_GROUP + 64 + lane]);
    float mh = fmaxf(o.err6h[(int64_t)grp * I8_GROUP + lane], o.err6h[(int64_t)grp * I8_GROUP + 64 + lane]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off));
        mh = fmaxf(mh, __shfl_xor(mh, off));
    }
    if (lane == 0) {
        o.gerr6[grp] = o.mx6_half ? __builtin_inff() : m;   // (a group of padding rows only: still "no full-width image")
        o.gerr6h[grp] = mh;
        o.gstep6[grp] = MX6_FIX_STEP;
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

}  // namespace

inline PrepOut prep_out(const Prepared& p, const int* perm = nullptr, int mx6_half = 0) {
    return PrepOut{p.inv, p.tiles, p.err, p.gstep, p.gerr, p.tiles8, p.tiles8h, p.rest, p.grest, perm, p.tiles6, p.err6, p.gerr6, p.gstep6,
                   p.err6h, p.gerr6h, p.rows8, mx6_half};
}

// Workgroups of prep_chunk_kernel (vfm_debug_set_prep_grid): -1 (default) = one per 128-row group; 0 = one per compute unit,
// each walking ceil(groups / grid) groups with the next group's rows read under the current group's quantisation and store
// (the kernel's registers admit one workgroup per compute unit, so nothing else overlaps them); n > 0 = n workgroups.
// Alone on the GPU the persistent form is the faster one (C2: 81 vs 109 us, 5.2 vs 3.9 TB/s); beside the coarse kernel of the
// previous registration -- whose workgroups need a whole compute unit each -- the short workgroups interleave better
// (791 vs 783 registrations/s over 300 steps, same box), and that is where the pipeline runs it.
inline int prep_grid(int groups, int mode) {
    static thread_local int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    const int knob = mode == VFM_PREPARE_DEFAULT ? vfm_cfg().prep_grid : (mode == VFM_PREPARE_PERSISTENT ? 0 : -1);
    int g = knob > 0 ? knob : (knob < 0 ? groups : cus);
    return g < groups ? g : groups;
}

// one or two operands (x2 may be NULL) in one launch.  want_f16 = false: only the int8 image (d = 256, 384), for operands that
// will meet in an int8 search (use_i8): a third of the bytes written, a third of the LDS.
int do_prepare2(Rows x1r, int64_t rows1, void* prepared1, Rows x2r, int64_t rows2, void* prepared2, int d,
                hipStream_t st, bool want_f16, int grid_mode) {
    // fp16 rows (VFM_ROWS_F16): the kernels that widen on load are prep_chunk_kernel and prep_mx6_rows_kernel -- the int8 and fp6
    // images of the gated family; the fp16-tile image (ungated fp16 pass) and the streamed fp6 form read fp32 rows only
    const bool any_f16 = x1r.f16 || (x2r.p && x2r.f16);
    if (any_f16 && (want_f16 || !i8_capable(d)))
        return vfm_fail(VFM_EINVAL, "prepare: fp16 rows are taken by the gated int8 / fp6 searches (d in {256 .. 768}, more queries than the fp16 pass' limit)");
    const float* x1 = static_cast<const float*>(x1r.p);   // (read as fp32 only where !any_f16)
    const float* x2 = static_cast<const float*>(x2r.p);
    const int h6 = (grid_mode & VFM_PREPARE_MX6_HALF) != 0 ? 1 : 0;   // the fp6 image of the first d / 2 columns only (implies VFM_PREPARE_MX6)
    if (h6) grid_mode |= VFM_PREPARE_MX6;
    const bool want_mx6 = (grid_mode & VFM_PREPARE_MX6) != 0 && mx6_width(d);              // int8 + fp6 image from one kernel
    const bool want_mx6_wide = (grid_mode & VFM_PREPARE_MX6) != 0 && !want_mx6 && mx6_half_width(d);   // d = 512, 768: kernels of their own
    grid_mode &= ~(VFM_PREPARE_MX6 | VFM_PREPARE_MX6_HALF);
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
            VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prep_chunk_kernel<false, 2, true, 8>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, I8_GROUP * (384 * 3 + 384 / 8 + 16)));
            attr_mark(attr_set);
        }
        const int groups = g1 + g2;
        int pg = prep_grid(groups, grid_mode);
        const dim3 grid((unsigned)pg), block(1024);
        if (want_mx6) {   // int8 + fp6 images from one read of the rows; the fp16 image, if wanted, by its own kernel
            // Which form: prep_stream_kernel was built to run BESIDE a coarse workgroup of round 4 (332 of a SIMD's 512 registers, 90 KiB).
            // Since round 5 the d = 384 coarse kernel holds three query tiles per wave (444 registers) and the full-width ones fill the
            // LDS: nothing runs beside them, and in a LONG pipeline the one-pass form below (rows in registers: 0.185 ms alone against
            // 0.139, but short fat workgroups that leave the coarse kernel alone) is the better neighbour -- tools/ab_prep_r5.py, 200-step
            // pipelines on one box: headline + 1.7 %, full width with the fused epilogue + 3.5 %, lifted descriptors + 2.2 %, full width
            // with records - 0.5 %.  In the driver's 20-step form it LOSES 2.9 % (tools/ab_prep_r5_20.py: 1624 against 1672
            // registrations/s): the default stays the stream form; vfm_debug_set_coarse_variant(42) selects by width (d = 384: one pass).
            const bool stream_form = vfm_cfg().prep_stream == 1 || (vfm_cfg().prep_stream == 2 && d == 256);
            if (vfm_cfg().prep_stream == 3 && (d == 384 || d == 256) && !any_f16) {   // one read of the rows, the fp16 copy of a tile in registers (prep_once_kernel, round 6)
                const dim3 sg((unsigned)groups), sb(256);
                if (d == 384 && h6) hipLaunchKernelGGL((prep_once_kernel<384, true>), sg, sb, 0, st, x1, rows1, prep_out(p1, nullptr, h6), g1, x2, rows2, prep_out(p2, nullptr, h6), groups);
                else if (d == 384) hipLaunchKernelGGL((prep_once_kernel<384, false>), sg, sb, 0, st, x1, rows1, prep_out(p1, nullptr, h6), g1, x2, rows2, prep_out(p2, nullptr, h6), groups);
                else if (h6) hipLaunchKernelGGL((prep_once_kernel<256, true>), sg, sb, 0, st, x1, rows1, prep_out(p1, nullptr, h6), g1, x2, rows2, prep_out(p2, nullptr, h6), groups);
                else hipLaunchKernelGGL((prep_once_kernel<256, false>), sg, sb, 0, st, x1, rows1, prep_out(p1, nullptr, h6), g1, x2, rows2, prep_out(p2, nullptr, h6), groups);
            } else
            if (stream_form && (d == 384 || d == 256) && !any_f16) {   // the form that fits beside a coarse workgroup (prep_stream_kernel)
                const dim3 sg((unsigned)groups), sb(256);
                if (d == 384 && h6) hipLaunchKernelGGL((prep_stream_kernel<384, true>), sg, sb, 0, st, x1, rows1, prep_out(p1, nullptr, h6), g1, x2, rows2, prep_out(p2, nullptr, h6), groups);
                else if (d == 384) hipLaunchKernelGGL((prep_stream_kernel<384, false>), sg, sb, 0, st, x1, rows1, prep_out(p1, nullptr, h6), g1, x2, rows2, prep_out(p2, nullptr, h6), groups);
                else if (h6) hipLaunchKernelGGL((prep_stream_kernel<256, true>), sg, sb, 0, st, x1, rows1, prep_out(p1, nullptr, h6), g1, x2, rows2, prep_out(p2, nullptr, h6), groups);
                else hipLaunchKernelGGL((prep_stream_kernel<256, false>), sg, sb, 0, st, x1, rows1, prep_out(p1, nullptr, h6), g1, x2, rows2, prep_out(p2, nullptr, h6), groups);
            } else
            hipLaunchKernelGGL((prep_chunk_kernel<false, 2, true, 8>), grid, dim3(512), (size_t)I8_GROUP * (d * 3 + d / 8 + 16), st, x1r, rows1, d,
                               prep_out(p1, nullptr, h6), g1, x2r, rows2, prep_out(p2, nullptr, h6), groups);
            if (want_f16)
                hipLaunchKernelGGL(prep_rows_kernel, dim3((unsigned)(t1 + t2)), dim3(256), (size_t)d * 64, st, x1, rows1, d, p1.inv,
                                   p1.tiles, t1, x2, rows2, p2.inv, p2.tiles);
        } else if (want_f16 && d <= 384) {  // both images from one read of the rows (144 KB of LDS at d = 384)
            hipLaunchKernelGGL((prep_chunk_kernel<true, 2>), grid, block, (size_t)I8_GROUP * d * 3, st, x1r, rows1, d, prep_out(p1), g1, x2r,
                               rows2, prep_out(p2), groups);
        } else {
            if (d <= 512)
                hipLaunchKernelGGL((prep_chunk_kernel<false, 2>), grid, block, (size_t)I8_GROUP * d, st, x1r, rows1, d, prep_out(p1), g1, x2r,
                                   rows2, prep_out(p2), groups);
            else
                hipLaunchKernelGGL((prep_chunk_kernel<false, 3>), grid, block, (size_t)I8_GROUP * d, st, x1r, rows1, d, prep_out(p1), g1, x2r,
                                   rows2, prep_out(p2), groups);
            if (want_f16) {  // wider rows: the fp16 image by its own kernel (both images would not fit the LDS)
                hipLaunchKernelGGL(prep_rows_kernel, dim3((unsigned)(t1 + t2)), dim3(256), (size_t)d * 64, st, x1, rows1, d, p1.inv,
                                   p1.tiles, t1, x2, rows2, p2.inv, p2.tiles);
            }
        }
        VFM_CHECK_LAUNCH("prep_chunk_kernel");
        if (want_mx6_wide) {
            const int64_t pad1 = rows_padded(rows1), total = pad1 + (x2 ? rows_padded(rows2) : 0);
            if (d == 512)
                hipLaunchKernelGGL((prep_mx6_rows_kernel<16>), dim3((unsigned)((total + 15) / 16)), dim3(256), (size_t)16 * d * 4, st, x1r, rows1, pad1, d,
                                   prep_out(p1, nullptr, h6), x2r, rows2, prep_out(p2, nullptr, h6), total);
            else
                hipLaunchKernelGGL((prep_mx6_rows_kernel<32>), dim3((unsigned)((total + 7) / 8)), dim3(256), (size_t)8 * d * 4, st, x1r, rows1, pad1, d,
                                   prep_out(p1, nullptr, h6), x2r, rows2, prep_out(p2, nullptr, h6), total);
            hipLaunchKernelGGL(prep_mx6_group_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, st, prep_out(p1, nullptr, h6), g1,
                               prep_out(p2, nullptr, h6), groups);
            VFM_CHECK_LAUNCH("prep_mx6_rows_kernel");
        }
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

int do_prepare_perm(const float* x, int64_t rows, const int* perm, int d, void* prepared, hipStream_t st) {
    if (!i8_capable(d)) return vfm_fail(VFM_EINVAL, "prepare(perm): no int8 image for d = %d", d);
    Prepared p1 = carve_prepared(prepared, rows, d);
    const int g1 = (int)(rows_padded(rows) / I8_GROUP);
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prep_chunk_kernel<false, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, I8_GROUP * 512));
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prep_chunk_kernel<false, 3>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, I8_GROUP * 768));
        attr_mark(attr_set);
    }
    const dim3 grid((unsigned)prep_grid(g1, VFM_PREPARE_DEFAULT)), block(1024);
    if (d <= 512)
        hipLaunchKernelGGL((prep_chunk_kernel<false, 2>), grid, block, (size_t)I8_GROUP * d, st, x, rows, d, prep_out(p1, perm), g1,
                           (const float*)nullptr, (int64_t)0, prep_out(Prepared{}), g1);
    else
        hipLaunchKernelGGL((prep_chunk_kernel<false, 3>), grid, block, (size_t)I8_GROUP * d, st, x, rows, d, prep_out(p1, perm), g1,
                           (const float*)nullptr, (int64_t)0, prep_out(Prepared{}), g1);
    VFM_CHECK_LAUNCH("prep_chunk_kernel(perm)");
    return VFM_OK;
}

}  // namespace vfmm

using namespace vfmm;

VFM_EXPORT int vfm_l2norm_rows_f32(float* x, int64_t n, int d, float* inv_out, vfm_stream_t stream) {
    VFM_CHECK_ARG(n >= 0 && d > 0 && d % 4 == 0 && d <= 1024, "l2norm: need d %% 4 == 0 and d <= 1024 (d=%d)", d);
    if (n == 0) return VFM_OK;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, n, d,
                       inv_out);
    VFM_CHECK_LAUNCH("l2norm_rows_kernel");
    return VFM_OK;
}

