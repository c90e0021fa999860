// match_coarse_mx6.hip -- the coarse pass of the matcher in microscaled fp6 (VFM_RECORDS_MX6, _MX6_TOP2, _MX6_HALF): gfx950's scaled MFMA
// v_mfma_scale_f32_32x32x64_f8f6f4 on e2m3 operands with one E8M0 scale per 32 columns does twice the int8 instruction's
// multiply-adds per cycle (32 cycles for 32 x 32 x 64; tools/probe/mx6_probe.hip: 6.4 PFLOP/s sustained on the whole chip
// against 4.4 for int8).  The fp6 image of prep_chunk_kernel<., ., true> has the int8 image's tile geometry -- two 1 KiB unit
// rows per k-step of 64 columns: 24 bytes of codes per lane (+ the lane's block scales in the spare bytes) -- so this is
// match_coarse_i8q2_kernel's schedule (64 resident queries per wave, four tiles = one 128-row chunk per step, ring of three
// steps staged by LDS-DMA, one barrier per step) with half as many MFMAs per tile.
//
// What comes out are the same best-score records: the chunk's best fp32 score x is written as ceil(x 2^20) + 2^30, an
// "integer score" with steps 2^-10 x 2^-10, so the selection kernels of the int8 pass read it unchanged with the fp6 image's
// residual norms as E (mx6_bounds).  The bound is the int8 pass's: with v = v^ + e (v^ the dequantised row, |e|_2 <= E measured),
//     | v_a . v_b - v^_a . v^_b |  <=  (|v_a| + E_a) E_b + E_a |v_b| ,
// the accumulation of v^_a . v^_b in fp32 (products of two e2m3 values and two powers of two are exact; six MFMA steps of at
// most a few ulp(4) each) and the 2^-20 grid of the record are inside MX6_SLACK, which prep adds to every E.
// Everything behind the selection -- int8 rescan of the candidate chunks, fp32 refinement, fp64 decision -- is the int8
// pass's and reads the int8 image.
#include "match_internal.h"

namespace vfmm {
namespace {

typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// one lane's operand of a k-step: 32 e2m3 codes in six registers (the instruction reads v[n:n+5] for fp6).  The block scales of a
// lane's KS6 k-steps sit together in bytes 8 .. 8 + KS6 - 1 of the lane's second unit of k-step 0 (unit row 1): two registers
// per tile and operand, the byte picked by the instruction's op_sel.
struct Mx6Frag {
    int c[6];
};
__device__ __forceinline__ Mx6Frag mx6_frag(const uint4* unit_row_a, int s) {   // unit rows 2 s, 2 s + 1 of the tile at `unit_row_a - lane`
    const uint4 lo = unit_row_a[(2 * s) * 64];
    const uint2 hi = *reinterpret_cast<const uint2*>(unit_row_a + (2 * s + 1) * 64);
    return Mx6Frag{{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y}};
}
__device__ __forceinline__ uint2 mx6_scales(const uint4* unit_row_a) {
    return *(reinterpret_cast<const uint2*>(unit_row_a + 64) + 1);
}
template <int S>
__device__ __forceinline__ floatx16 mfma_mx6(const Mx6Frag& x, const uint2& xs, const Mx6Frag& y, const uint2& ys, floatx16 c) {
    intx8 a, b;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        a[i] = x.c[i];
        b[i] = y.c[i];
    }
    a[6] = a[7] = b[6] = b[7] = 0;   // not read: cbsz = blgp = 2 (e2m3) takes six registers per operand
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, S & 3, (int)(S < 4 ? xs.x : xs.y), S & 3,
                                                           (int)(S < 4 ? ys.x : ys.y));
}

// TOP2: packed top-2 records (VFM_RECORDS_MX6_TOP2).  The accumulators then start at 2.0 (a splat the compiler keeps in sixteen registers: unlike 0 it is not folded into the
// instruction): scores of unit
// rows stay in [0.9, 3.1], positive floats whose bit patterns order like the values, and coarse_fold -- the fp16 pass's: low six
// bits replaced by the row code, running best / second best by max / med3 -- works on them as it stands.  At the end of a
// chunk the two packed values become fixed-point integers, each rounded UP from the top of its packing interval (64 ulp of 4 =
// 1.5e-5): the selection reads record | 127 / | 63 as upper bounds, and the lower bound (record & ~127) can exceed the truth by
// at most that interval, which MX6_SLACK covers.
// IMG_KS6 > KS6 (VFM_RECORDS_MX6_HALF): the pass runs over the first KS6 k-steps of an image whose tiles hold IMG_KS6 -- the
// half-width pass in fp6.  A tile's first 2 KS6 unit rows are a prefix of the stored tile, so there is no second image: the
// staging packs them into the ring unit row by unit row (wave w copies packed rows w, w + 8, ... of a step; packed row r is
// unit row r % UNITS of tile r / UNITS), the queries' registers take the first KS6 k-steps, the scales sit in unit row 1 as ever.
template <int KS6, bool TOP2 = false, bool LOW = true, int IMG_KS6 = KS6>
__global__ __launch_bounds__(512, 2) void match_coarse_mx6q2_kernel(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVES = 8, T = 4;
    constexpr int UNITS = 2 * KS6;              // 1 KiB unit rows per tile
    constexpr int TILE_U4 = UNITS * 64;
    constexpr int TILE_BYTES = TILE_U4 * 16;
    constexpr int PIECES = T * UNITS / NWAVES;  // 1 KiB pieces per wave per step: 6 (d = 384) or 4 (d = 256)
    constexpr int NBUF = 3 * T;
    constexpr int IMG_TILE_U4 = 2 * IMG_KS6 * 64;   // stride of the stored tiles
    constexpr int PF = KS6 % 3 == 0 ? 3 : 2;    // fragment look-ahead in k-steps (PF == KS6: every read is of the next tile)
    static_assert(KS6 % PF == 0 && (KS6 >= 2 * PF || KS6 == PF) && PIECES <= 2 * T && (T * UNITS) % NWAVES == 0 && KS6 >= 2 && KS6 <= 6 &&
                      IMG_KS6 >= KS6 && IMG_KS6 <= 12, "shape (the scales of k-steps 0 .. 7 sit in unit row 1: KS6 <= 8)");
    static_assert(NBUF * TILE_BYTES <= 160 * 1024, "ring exceeds the LDS");
    // One barrier per step, between tiles 2 and 3.  A step of this kernel is half as long as the int8 kernel's for the same bytes
    // staged, so the staging loads need more of it: the pieces that lie inside tiles 0-2 of their step (EARLY of them) are
    // issued during tile 0 -- their ring slot was last read before the previous step's barrier -- the others during tile 3;
    // a wave waits for everything but its newest EARLY loads right before the barrier, and behind it tile 3's look-ahead
    // reads the first fragments of the next step.  Every piece has at least a whole step to land (the int8 kernel: 10 of 24
    // k-steps for the last one).
    constexpr int EARLY = 3 * UNITS / 8, LATE = PIECES - EARLY;
    static_assert(EARLY >= 1 && LATE >= 1 && EARLY <= KS6 && LATE <= KS6, "piece schedule");

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const CoarseUnit cu = coarse_unit(a);
    const int qb = cu.qb, c0 = cu.c0, ntiles = cu.ntiles;  // ntiles: a multiple of 4 (whole chunks)
    if (ntiles == 0) return;
    const int qt0 = (qb * NWAVES + wave) * 2;  // this wave's two 32-query tiles

    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS unsigned char*)smem;
    const uint4* gsrc = a.Bh + (size_t)c0 * 4 * IMG_TILE_U4 + lane;   // the step's first stored tile
    const unsigned ldst0 = lds_base + (unsigned)wave * 1024u;
    // piece p of a step for this wave: packed unit row wave + 8 p = unit row r % UNITS of tile r / UNITS (contiguous when
    // IMG_KS6 == KS6)
    auto piece_src = [&](const uint4* step_base, int p) __attribute__((always_inline)) {
        const int r = wave + NWAVES * p;
        return step_base + (r / UNITS) * IMG_TILE_U4 + (r % UNITS) * 64;
    };
    auto stage_step = [&](const uint4* src, unsigned ring_byte) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            glds16(piece_src(src, p), __builtin_amdgcn_readfirstlane(ldst0 + ring_byte + (unsigned)(p * NWAVES) * 1024u));
    };

    Mx6Frag qf[2][KS6];
    uint2 qs[2];
    float fx_sq[2], fx_A[2], fx_mult[2], fx_low[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qt = qt0 + j < a.nq_tiles ? qt0 + j : 0;
        const uint4* qsrc = a.Qh + (size_t)qt * IMG_TILE_U4 + lane;
#pragma unroll
        for (int s = 0; s < KS6; ++s) qf[j][s] = mx6_frag(qsrc, s);
        qs[j] = mx6_scales(qsrc);
        const size_t qi = (size_t)qt * 32 + (lane & 31);
        const float eq = a.ib.qerr[qi];
        fx_sq[j] = a.ib.qstep[qi >> 7];
        fx_A[j] = eq * 1.0001220703125f + 1.0e-6f;
        fx_mult[j] = 1.0001220703125f + eq;
        fx_low[j] = -__builtin_inff();
    }
    stage_step(gsrc, 0u);
    if (ntiles > T) stage_step(gsrc + (size_t)T * IMG_TILE_U4, (unsigned)(T * TILE_BYTES));
    const uint4* gnext = gsrc + (size_t)2 * T * IMG_TILE_U4;

    float s1[2] = {-__builtin_inff(), -__builtin_inff()};   // the scores are floats: v_max3_f32 folds them as they are
    unsigned t1[2] = {0u, 0u}, t2[2] = {0u, 0u};            // TOP2: running best / second best, packed (coarse_fold)
    constexpr float ACC0 = TOP2 ? 2.0f : 0.0f;
    auto emit_chunk = [&](int chunk) __attribute__((always_inline)) {  // chunk < 0: nothing folded yet
        float sb = 0.f, be = 0.f;
        const bool counted = LOW && chunk >= 0;  // wave-uniform
        if (counted) {
            sb = a.ib.bstep[chunk];
            be = a.ib.berr[chunk];
        }
        const bool padded = chunk >= a.first_pad_chunk;   // zero-padded rows score exactly 0
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned best;
            if constexpr (TOP2) {
                // merge the two half-waves (coarse_emit_chunk's rule: the larger packed value, ties to the lower half), then to
                // fixed point
                const int lane_ = lane_id(), hi = lane_ >> 5;
                const unsigned o1 = __shfl_xor(t1[j], 32), o2 = __shfl_xor(t2[j], 32);
                const bool own = (t1[j] > o1) || (t1[j] == o1 && hi == 0);
                const unsigned w1 = own ? t1[j] : o1;
                const int wh = own ? hi : (1 - hi);
                const unsigned w2 = max(max(t2[j], o2), min(t1[j], o1));
                const int code = 63 - (int)(w1 & 63u);
                const int li = (code >> 4) * 32 + (code & 3) + 8 * ((code & 15) >> 2) + 4 * wh;  // row inside the chunk
                const int f1 = (int)ceilf((fmaxf(__uint_as_float(w1 | 63u), 0.0f) - 2.0f) * 1048576.0f) + I8_OFFSET;
                const int f2 = (int)ceilf((fmaxf(__uint_as_float(w2 | 63u), 0.0f) - 2.0f) * 1048576.0f) + I8_OFFSET;
                if (lane_ < 32 && qt0 + j < a.nq_tiles && chunk >= 0)
                    a.partials[((size_t)(qt0 + j) * a.nchunks + (size_t)chunk) * 32 + lane_] = make_uint2(((unsigned)f1 & ~127u) | (unsigned)li, (unsigned)f2);
                t1[j] = 0u;
                t2[j] = 0u;
                best = (unsigned)f1 & ~127u;
            } else {
                // the lane's best fp32 score of the chunk -> fixed point, rounded up
                const int fix = (int)ceilf(fmaxf(s1[j], -4.0f) * 1048576.0f);
                unsigned rec = (unsigned)(fix + I8_OFFSET);
                best = coarse_emit_chunk_best(a, rec, qt0 + j, chunk);
                s1[j] = -__builtin_inff();
            }
            if (counted) {
                const int sbest = (int)best - I8_OFFSET;
                const float low = __builtin_fmaf(fx_sq[j] * sb, (float)sbest, -(fx_A[j] + fx_mult[j] * be));
                fx_low[j] = fmaxf(fx_low[j], (!padded || sbest > 0) ? low : -__builtin_inff());
            }
        }
    };

    floatx16 accA[2], accB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[j][r] = accB[j][r] = TOP2 ? 0.0f : -__builtin_inff();   // (folded by the first tile: no effect)

    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Mx6Frag fr[PF];  // fragment ring registers of the tile in progress: k-step s consumes slot s % PF
    uint2 sc_cur, sc_nxt;   // block scales of the tile in progress / of the tile after it
    {
        const uint4* b0 = reinterpret_cast<const uint4*>(smem) + lane;
#pragma unroll
        for (int s = 0; s < PF; ++s) fr[s] = mx6_frag(b0, s);
        sc_nxt = mx6_scales(b0);
    }
    unsigned ring = 0u;

    for (int it = 0; it < ntiles; it += T) {
        const unsigned ring1 = ring + T >= (unsigned)NBUF ? ring + T - NBUF : ring + T;
        const unsigned ring2 = ring1 + T >= (unsigned)NBUF ? ring1 + T - NBUF : ring1 + T;
        const uint4* cur = reinterpret_cast<const uint4*>(smem + ring * TILE_BYTES) + lane;
        const uint4* nxt = reinterpret_cast<const uint4*>(smem + ring1 * TILE_BYTES) + lane;
        const bool more = it + 2 * T < ntiles;
        // tile J of the step into `acc`, folding `done` (the tile before it)
        auto tile = [&](auto Jc, floatx16 (&acc)[2], const floatx16 (&done)[2]) __attribute__((always_inline)) {
            constexpr int J = decltype(Jc)::value;
            const uint4* tb = cur + J * TILE_U4;
            const uint4* tn = (J + 1 < T) ? cur + (J + 1) * TILE_U4 : nxt;  // the tile after it (stale after the last step)
            sc_cur = sc_nxt;
            sc_nxt = mx6_scales(tn);
            auto kstep = [&](auto Sc) __attribute__((always_inline)) {
                constexpr int s = decltype(Sc)::value;
                if constexpr (s == 0) {   // the tile's first MFMAs start from an inline constant (0, or 2.0): no accumulator to clear
                    floatx16 zero;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zero[r] = ACC0;
                    acc[0] = mfma_mx6<s>(fr[s % PF], sc_cur, qf[0][s], qs[0], zero);
                    acc[1] = mfma_mx6<s>(fr[s % PF], sc_cur, qf[1][s], qs[1], zero);
                } else {
                    acc[0] = mfma_mx6<s>(fr[s % PF], sc_cur, qf[0][s], qs[0], acc[0]);
                    acc[1] = mfma_mx6<s>(fr[s % PF], sc_cur, qf[1][s], qs[1], acc[1]);
                }
                // (an empty use of the results: without it the MFMAs -- pure functions to the compiler -- are sunk across the
                // blocks of the step to their first reader, the fold one tile later, and every fragment of the step is read before
                // the first of them: 139 spilled registers)
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
                fr[s % PF] = (s + PF < KS6) ? mx6_frag(tb, s + PF) : mx6_frag(tn, s + PF - KS6);
#pragma unroll
                for (int e = s * 32 / KS6; e < (s + 1) * 32 / KS6; ++e) {   // `done` is tile (J + 3) & 3 of its chunk
                    if constexpr (TOP2) coarse_fold(t1[e >> 4], t2[e >> 4], done[e >> 4][e & 15], ((J + 3) & 3) * 16 + (e & 15));
                    else s1[e >> 4] = fmaxf(s1[e >> 4], done[e >> 4][e & 15]);
                }
                if ((J == 0 && s < EARLY) || (J == 3 && s < LATE)) {  // one 1 KiB piece per k-step
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        const int p = J == 0 ? s : EARLY + s;
                        glds16(piece_src(gnext, p),
                               __builtin_amdgcn_readfirstlane(ldst0 + ring2 * TILE_BYTES + (unsigned)(p * NWAVES) * 1024u));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
            if constexpr (KS6 > 2) kstep(std::integral_constant<int, 2>{});
            if constexpr (KS6 > 3) kstep(std::integral_constant<int, 3>{});
            if constexpr (KS6 > 4) {
                kstep(std::integral_constant<int, 4>{});
                kstep(std::integral_constant<int, 5>{});
            }
        };
        tile(std::integral_constant<int, 0>{}, accA, accB);
        emit_chunk(it >= 4 ? c0 + (it >> 2) - 1 : -1);  // tile 0's slots folded the last tile of the previous chunk
        tile(std::integral_constant<int, 1>{}, accB, accA);
        tile(std::integral_constant<int, 2>{}, accA, accB);
        // the next step's pieces (issued during the previous step) have landed; this step's early ones -- the newest EARLY loads,
        // if it issued any -- may fly on
        if (more) wait_vmcnt<EARLY>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tile(std::integral_constant<int, 3>{}, accB, accA);
        gnext += T * IMG_TILE_U4;
        ring = ring1;
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        if constexpr (TOP2) coarse_fold(t1[e >> 4], t2[e >> 4], accB[e >> 4][e & 15], 3 * 16 + (e & 15));
        else s1[e >> 4] = fmaxf(s1[e >> 4], accB[e >> 4][e & 15]);
    }
    emit_chunk(c0 + (ntiles >> 2) - 1);
    if constexpr (LOW) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (lane < 32 && qt0 + j < a.nq_tiles) atomicMax(a.qmax + (size_t)(qt0 + j) * 32 + lane, float_key(fx_low[j]));
    }
}

template <int KS6, bool TOP2>
int launch_mx6q2(const CoarseArgs& a, hipStream_t st) {
    const int lds = 12 * (2 * KS6) * 1024;
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_coarse_mx6q2_kernel<KS6, TOP2, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_mark(attr_set);
    }
    hipLaunchKernelGGL((match_coarse_mx6q2_kernel<KS6, TOP2, true>), dim3(a.nqb * a.nslices), dim3(512), lds, st, a);
    return VFM_OK;
}

// the half-width pass in fp6: the first d / 2 columns of the same image, best-score records, no running lower bound
template <int KS6, int IMG_KS6>
int launch_mx6q2_half(const CoarseArgs& a, hipStream_t st) {
    const int lds = 12 * (2 * KS6) * 1024;
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_coarse_mx6q2_kernel<KS6, false, false, IMG_KS6>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_mark(attr_set);
    }
    hipLaunchKernelGGL((match_coarse_mx6q2_kernel<KS6, false, false, IMG_KS6>), dim3(a.nqb * a.nslices), dim3(512), lds, st, a);
    return VFM_OK;
}

}  // namespace

// the fp6 coarse kernel for the arguments do_search_coarse prepared (a.Qh / a.Bh = the fp6 tiles, a.ib = mx6_bounds);
// d = 256 / 384 and more than 2048 queries (effective_records)
int launch_coarse_mx6(CoarseArgs& a, int d, bool top2, bool half, hipStream_t st) {
    a.nqb = (a.nq_tiles + 15) / 16;
    a.nslices = choose_slices(a.nqb, a.nchunks);
    if (half && g_force_slices == 0) {
        // the half-width kernel's workgroups are short, and shorter ones let the other stages of a pipeline in: ~7.5 rounds of 256
        // workgroups instead of choose_slices' 5 (tools/sweep_slices.py, C2, 200 steps: 48 slices 1548 against 32 slices 1475
        // registrations/s; the full-width fp6 kernel and the int8 kernels are flat from 32 on)
        int s = (int)((1920 + a.nqb - 1) / a.nqb);
        const int smax = a.nchunks / 8 < 64 ? a.nchunks / 8 : 64;
        s = s > smax ? smax : s;
        a.nslices = s < 1 ? 1 : s;
    }
    if (g_prof_start) VFM_CHECK_HIP(hipEventRecord(g_prof_start, st));
    int rc;
    if (half) rc = d == 768 ? launch_mx6q2_half<6, 12>(a, st) : d == 512 ? launch_mx6q2_half<4, 8>(a, st)
                            : d == 384 ? launch_mx6q2_half<3, 6>(a, st) : launch_mx6q2_half<2, 4>(a, st);
    else rc = d == 384 ? (top2 ? launch_mx6q2<6, true>(a, st) : launch_mx6q2<6, false>(a, st))
                       : (top2 ? launch_mx6q2<4, true>(a, st) : launch_mx6q2<4, false>(a, st));
    if (rc) return rc;
    VFM_CHECK_LAUNCH("match_coarse_mx6q2_kernel");
    if (g_prof_stop) VFM_CHECK_HIP(hipEventRecord(g_prof_stop, st));
    g_prof_start = g_prof_stop = nullptr;
    return VFM_OK;
}

}  // namespace vfmm
