// match_coarse_mx6.hip -- the coarse pass of the matcher in microscaled fp6 (VFM_RECORDS_MX6, _MX6_TOP2, _MX6_HALF, _MX6_HALF_FUSED):
// gfx950's scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4 on e2m3 operands with one E8M0 scale per 32 columns does twice the int8
// instruction's multiply-adds per cycle (32 cycles for 32 x 32 x 64; tools/probe/mx6_probe.hip: 6.4 PFLOP/s sustained on the
// whole chip against 4.4 for int8).
//
// Round 4 rework.  What the counters said about round 3's kernel: 35 % of its LDS cycles were bank
// conflicts (a lane's 24 code bytes were read as ds_read_b128 + ds_read_b64 from 16-byte units: the 8-byte reads of a 32-lane
// group at 16-byte stride cover each bank twice), 31 % of the wave time was spent parked in s_waitcnt / s_barrier, and the
// compiler had sunk the fold of three tiles into one block of 45 dependent v_max3 behind the barrier, where both waves of a SIMD
// run it at the same time with the matrix pipe idle.  Now:
//   * the image is DENSE (match_internal.h, mx6_*): per tile a 512-byte plane of scales, then per k-step a 1 KiB plane of the
//     lanes' first 16 code bytes and a 512-byte plane of their last 8 -- every fragment read is conflict-free (ds_read_b128 at
//     16-byte stride, ds_read_b64 at 8-byte stride), a tile is 9.5 KiB instead of 12 (d = 384), and the half-width pass reads a
//     5 KiB prefix of it: a quarter less L2 -> LDS staging and LDS footprint;
//   * the ring is RING >= 4 steps deep: a step's pieces are issued right behind the barrier that frees their slot and have at
//     least one whole step beyond the one they were issued in to land (s_waitcnt vmcnt((RING - 3) x pieces));
//   * nothing in the loop waits for lgkmcnt(0): the half-waves are merged by v_permlane32_swap (a ds_bpermute drained the
//     fragment look-ahead once per chunk and query set), the per-chunk constants of the bounds come from an LDS table read a
//     chunk ahead (scalar loads share lgkmcnt with the LDS);
//   * the fold of a finished tile is pinned into the k-step slots of the next one (empty asm uses of the running maxima);
//   * FUSE (VFM_RECORDS_MX6_HALF_FUSED): the half-width pass writes NO records.  The chunk's best score is tested against the
//     gate in registers -- the bound of match_select_half_kernel -- and a survivor (0.56 per query on SURVEY D.2 data) is
//     appended to a list in the LDS; the workgroup leaves its list in a slot of its own in global memory (plain stores: no
//     atomic, nothing to wait for) and match_bin_survivors_kernel (match_finish.hip) bins the ~11 000 entries per chunk.
//     The 122 MB record array, its 70 us sweep (match_select_half_kernel) and the global atomics of round 2's fused int8 form
//     are gone.  A workgroup whose list overflows (descriptors that are all alike) raises the device-side guard of the
//     half-width pass and stops recording: match_gatepass_kernel then decides every query.
//
// What comes out of the record kinds is unchanged: the chunk's best fp32 score x as ceil(x 2^20) + 2^30, an "integer score" with
// steps 2^-10 x 2^-10, read by the selection kernels of the int8 pass with the fp6 image's residual norms as E (mx6_bounds).
// The bound is the int8 pass's: with v = v^ + e (v^ the dequantised row, |e|_2 <= E measured),
//     | v_a . v_b - v^_a . v^_b |  <=  (|v_a| + E_a) E_b + E_a |v_b| ,
// the accumulation of v^_a . v^_b in fp32 (products of two e2m3 values and two powers of two are exact; six MFMA steps of at
// most a few ulp(4) each) and the 2^-20 grid of the record are inside MX6_SLACK, which prep adds to every E.
// Everything behind the selection -- int8 rescan of the candidate chunks, fp32 refinement, fp64 decision -- is the int8
// pass's and reads the int8 image.
#include "match_internal.h"

#include <atomic>

namespace vfmm {
namespace {

typedef int intx8 __attribute__((ext_vector_type(8)));

// one lane's operand of a k-step: 32 e2m3 codes in six registers (the instruction reads v[n:n+5] for fp6)
struct Mx6Frag {
    int c[6];
};
// k-step s of the tile whose code planes start at pa (this lane's 16-byte slot of plane A) / pb (its 8-byte slot of plane B)
__device__ __forceinline__ Mx6Frag mx6_frag(const unsigned char* pa, const unsigned char* pb, int s) {
    const uint4 lo = *reinterpret_cast<const uint4*>(pa + s * MX6_KSTEP_BYTES);
    const uint2 hi = *reinterpret_cast<const uint2*>(pb + s * MX6_KSTEP_BYTES);
    return Mx6Frag{{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y}};
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
#ifdef VFM_ABL_NOSCALE   // (timing experiment, tools/ablate6.py: the unscaled instruction -- results are garbage)
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, S & 3, (int)(S < 4 ? xs.x : xs.y), S & 3,
                                                           (int)(S < 4 ? ys.x : ys.y));
#endif
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// LDS-DMA of 16 bytes per lane from `base` + this lane's byte offset (saddr + 32-bit voffset form) to LDS address m0 + 16 lane
// (m0 is saved and restored: the compiler does not track it across inline asm)
__device__ __forceinline__ void glds16s(const void* base_uniform, unsigned voff, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base_uniform), "s"(lds_dst_uniform)
        : "memory");
}
// the other half-wave's value (lanes l and l ^ 32 exchange) by v_permlane32_swap: VALU, no LDS counter involved.  With vdst = src =
// v the instruction leaves {own, other} in lanes 0 - 31 and {other, own} in lanes 32 - 63 of its two results.
__device__ __forceinline__ unsigned swap32(unsigned v) {
    const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return lane_id() < 32 ? (unsigned)sw[1] : (unsigned)sw[0];
}
__device__ __forceinline__ unsigned umax_halves(unsigned v) {
    const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return max((unsigned)sw[0], (unsigned)sw[1]);
}
__device__ __forceinline__ float fmax_halves(float v) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float((unsigned)sw[0]), __uint_as_float((unsigned)sw[1]));
}

enum { MX6_BEST = 0, MX6_TOP2 = 1, MX6_FUSE = 2 };
constexpr int MX6_LTAB = 256;       // per-chunk constants of a workgroup's slice kept in the LDS (launch_coarse_mx6 keeps slices this short)
constexpr int MX6_LCAP = 2047;      // FUSE: survivors a workgroup can list (+ the header = 8 KiB)

// KIND: MX6_BEST best-score records, MX6_TOP2 packed top-2 records (VFM_RECORDS_MX6_TOP2: the accumulators then start at 2.0 --
// scores of unit rows stay in [0.9, 3.1], positive floats whose bit patterns order like the values, and coarse_fold -- the fp16
// pass's: low six bits replaced by the row code, running best / second best by max / med3 -- works on them as it stands; at the
// end of a chunk the two packed values become fixed-point integers, each rounded UP from the top of its packing interval),
// MX6_FUSE no records: survivors of the half-width bound (see the head of the file).
// IMG_KS6 > KS6 (the half-width pass): the pass runs over the first KS6 k-steps of an image whose tiles hold IMG_KS6; the scales
// and the first KS6 k-steps are a prefix of the stored tile.
// T = tiles per step = per barrier: 4 (one chunk; RING >= 4 steps) or 8 (two chunks, RING = 3: half as many barriers -- the ablations
// of tools/ablate6.py put barrier + staging at 10 % of the kernel -- for the shapes whose ring of 3 x 8 tiles fits the LDS)
// NS (round 5): 32-query tiles resident per wave -- 2, or 3 for the fused half-width pass at d = 384 (96 queries per wave, 768 per workgroup:
// every map fragment read from the LDS feeds three MFMAs instead of two, and a step between two barriers is 36 MFMAs per wave instead of 24;
// 54 query + 96 accumulator registers)
template <int KS6, int KIND, bool LOW, int IMG_KS6, int RING, int T = 4, int NS = 2>
__global__ __launch_bounds__(512, 2) void match_coarse_mx6q2_kernel(CoarseArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (The host pass only needs this kernel's stub.  With the accumulator sets sized by NS its semantic check of the body -- inline asm with
    // register constraints inside nested generic lambdas -- failed WITHOUT A DIAGNOSTIC (device-side bodies' errors are deferred on the
    // host) and the stub of every instantiation went missing: the library then failed to load with undefined kernel symbols.)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool TOP2 = KIND == MX6_TOP2, FUSE = KIND == MX6_FUSE;
    constexpr int NWAVES = 8;
    static_assert(T == 4 || T == 8, "a step is one or two chunks");
    constexpr int LT = MX6_SCALE_PLANE + KS6 * MX6_KSTEP_BYTES;   // bytes of a tile in the LDS (the prefix the pass reads)
    constexpr int IMG_TB = mx6_tile_bytes(IMG_KS6);               // bytes of a stored tile
    constexpr int STEP_BYTES = T * LT;
    constexpr int NP = STEP_BYTES / 1024;                         // 1 KiB pieces per step
    constexpr int PWMAX = (NP + NWAVES - 1) / NWAVES, PWFULL = NP % NWAVES;   // pieces per wave: PWMAX for waves < PWFULL (all, if 0), else one less
    constexpr int PF = KS6 % 3 == 0 ? 3 : 2;    // fragment look-ahead in k-steps (PF == KS6: every read is of the next tile)
    static_assert(STEP_BYTES % 1024 == 0 && KS6 % PF == 0 && (KS6 >= 2 * PF || KS6 == PF) && KS6 >= 2 && KS6 <= 6 && IMG_KS6 >= KS6 &&
                      IMG_KS6 <= 12 && PWMAX <= 3 * KS6 && (RING >= 4 || (RING == 3 && T == 8)),
                  "shape (the scales of k-steps 0 .. 7 sit in the first scale plane; a wave's pieces go out in k-step slots behind the barrier)");
    // FUSE: entries the workgroup's list holds (the slot in global memory is MX6_LCAP + 4 words for every shape; d = 768's ring leaves 4 KiB)
    constexpr int LCAP = RING * STEP_BYTES + MX6_LTAB * 8 + (MX6_LCAP + 1) * 4 <= 160 * 1024 ? MX6_LCAP : 1023;   // (= launch_mx6q2's)
    static_assert(RING * STEP_BYTES + MX6_LTAB * 8 + (FUSE ? (LCAP + 1) * 4 : 0) <= 160 * 1024, "ring exceeds the LDS");
    static_assert(!FUSE || !LOW, "the fused forms keep no running lower bound");
    static_assert(NS == 2 || (NS == 3 && FUSE && (16 * NS) % KS6 == 0), "three query sets: the fused forms whose fold divides evenly over the k-steps");
    constexpr bool FUSE_HALF = FUSE && IMG_KS6 > KS6;   // FUSE at full width (VFM_RECORDS_MX6_FUSED, round 5): the same test without the rest term

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    CoarseUnit cu = coarse_unit(a);
    if constexpr (T == 8) {   // whole PAIRS of chunks per unit (the padded map has an even number of chunks: rows_padded)
        const int total = a.nqb * a.nslices, bid = blockIdx.x;
        const int xcd = bid & 7, within = bid >> 3, qn = total >> 3, rn = total & 7;
        const int unit = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + within;
        const int slice = unit / a.nqb, npairs = a.nchunks >> 1;
        cu.qb = unit - slice * a.nqb;
        cu.c0 = 2 * (int)(((long long)slice * npairs) / a.nslices);
        cu.ntiles = (2 * (int)(((long long)(slice + 1) * npairs) / a.nslices) - cu.c0) * 4;
    }
    const int qb = cu.qb, c0 = cu.c0, ntiles = cu.ntiles;  // ntiles: a multiple of T (whole chunks / pairs of chunks)
    if (ntiles == 0) {
        if (KIND == MX6_FUSE && threadIdx.x == 0) a.surv[(size_t)blockIdx.x * (MX6_LCAP + 1 + 3)] = 0u;   // an empty slot
        return;
    }
    const int qt0 = (qb * NWAVES + wave) * NS;  // this wave's NS 32-query tiles
    float2* ltab = reinterpret_cast<float2*>(smem + RING * STEP_BYTES);
    unsigned* llist = reinterpret_cast<unsigned*>(smem + RING * STEP_BYTES + MX6_LTAB * 8);   // FUSE: [0] = count, then the entries

    // ---- staging: piece p of a step = bytes [1024 p, 1024 p + 1024) of the step's LDS image; wave w issues pieces w, w + 8, ...
    // A lane's 16 bytes of piece p sit at byte o = 1024 p + 16 lane of the image = byte o % LT of LDS tile o / LT = the same byte
    // of the stored tile (stride IMG_TB): the offsets are per-lane constants of the kernel.
    const unsigned lds_base = (unsigned)(uintptr_t)(LDS_AS unsigned char*)smem;
    const int my_pieces = (PWFULL == 0 || wave < PWFULL) ? PWMAX : PWMAX - 1;   // wave-uniform
    unsigned goff[PWMAX];
#pragma unroll
    for (int k = 0; k < PWMAX; ++k) {
        const int o = 1024 * (wave + NWAVES * k) + 16 * lane;
        goff[k] = (unsigned)((o / LT) * IMG_TB + (o % LT));
    }
    const unsigned char* gstep = reinterpret_cast<const unsigned char*>(a.Bh) + (size_t)c0 * 4 * IMG_TB;   // stored tiles of the step being staged (4 tiles per chunk)
    auto issue_piece = [&](int k, unsigned slot_byte) __attribute__((always_inline)) {
        glds16s(gstep, goff[k], __builtin_amdgcn_readfirstlane(lds_base + slot_byte + 1024u * (unsigned)(wave + NWAVES * k)));
    };
    auto stage_step_now = [&](unsigned slot_byte) {
#pragma unroll
        for (int k = 0; k < PWMAX; ++k)
            if (k < my_pieces) issue_piece(k, slot_byte);
    };

    // ---- the wave's queries (two 32-query tiles of the same image layout) and the per-query terms of the bounds
    Mx6Frag qf[NS][KS6];
    uint2 qs[NS];
    float fx_A[NS], fx_mult[NS], fx_low[NS], fx_rq[NS];
    int fx_arg[NS] = {};
    bool live[NS] = {};
    unsigned livemask[NS] = {};   // FUSE: the set's queries that exist and are not zero rows, one bit per query of the tile
    const unsigned char* qimg = reinterpret_cast<const unsigned char*>(a.Qh);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int qt = qt0 + j < a.nq_tiles ? qt0 + j : 0;
        const unsigned char* qtile = qimg + (size_t)qt * IMG_TB;
#pragma unroll
        for (int s = 0; s < KS6; ++s) qf[j][s] = mx6_frag(qtile + MX6_SCALE_PLANE + 16 * lane, qtile + MX6_SCALE_PLANE + 1024 + 8 * lane, s);
        qs[j] = *reinterpret_cast<const uint2*>(qtile + 8 * lane);
        const size_t qi = (size_t)qt * 32 + (lane & 31);
        const float eq = a.ib.qerr[qi];
        fx_A[j] = eq * 1.0001220703125f + 1.0e-6f;
        fx_mult[j] = 1.0001220703125f + eq;
        fx_low[j] = -__builtin_inff();
        fx_rq[j] = 0.0f;
        if constexpr (FUSE) {
            if constexpr (FUSE_HALF) fx_rq[j] = a.qrest[qi];
            live[j] = lane < 32 && qt0 + j < a.nq_tiles && (int64_t)qi < a.n_valid && a.qinv[qi] != 0.0f;
            livemask[j] = (unsigned)__ballot(live[j]);
        }
    }
    // per-chunk constants of the slice: (step, max E) for the running lower bound, (max E, max |rest|) for the fused test
    const int nch = ntiles >> 2;
    for (int c = threadIdx.x; c < nch; c += 512)
        ltab[c] = FUSE ? make_float2(a.ib.berr[c0 + c], FUSE_HALF ? a.grest[c0 + c] : 0.0f) : make_float2(a.ib.bstep[c0 + c], a.ib.berr[c0 + c]);
    if (FUSE && threadIdx.x == 0) llist[0] = 0u;

    // steps 0 .. RING - 2 of the unit (as many as it has) go out before the loop
#pragma unroll
    for (int r = 0; r < RING - 1; ++r) {
        if (r * T < ntiles) stage_step_now((unsigned)(r * STEP_BYTES));
        gstep += T * IMG_TB;
    }
    // (gstep now points at step RING - 1, the first one the loop issues -- behind the barrier of step 0, into the last slot)

    float s1[NS];                                           // the scores are floats: v_max3_f32 folds them as they are
    unsigned t1[NS] = {}, t2[NS] = {};                      // TOP2: running best / second best, packed (coarse_fold)
#pragma unroll
    for (int j = 0; j < NS; ++j) s1[j] = -__builtin_inff();
    constexpr float ACC0 = TOP2 ? 2.0f : 0.0f;
    unsigned seen = 0u;                                     // FUSE: the list's length as of the previous chunk (read a chunk ahead, like tab)
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    float2 tab = make_float2(0.f, 0.f);                     // ltab entry of the chunk emit_chunk sees next (read a chunk ahead)
    float thr[NS];                                          // FUSE: the score from which that chunk survives for the lane's query
#pragma unroll
    for (int j = 0; j < NS; ++j) thr[j] = __builtin_inff();
    // (from `tab`, in the last tile of the step in front of the emit: the emit itself -- both waves of a SIMD reach it together, a
    // barrier earlier they were released together -- then holds no chain of dependent VALU results in front of its branch)
    auto next_thr = [&]() __attribute__((always_inline)) {
        if constexpr (FUSE) {
#pragma unroll
            for (int j = 0; j < NS; ++j) thr[j] = ((a.gate - 1.0e-6f) - (fx_A[j] + fx_mult[j] * tab.x)) - (fx_rq[j] * tab.y + 1.0e-6f);
#pragma unroll
            for (int j = 0; j < NS; ++j) asm volatile("" ::"v"(thr[j]));
        }
    };
    auto emit_chunk = [&](int ci) __attribute__((always_inline)) {  // ci = chunk of the unit, < 0: nothing folded yet
        const bool valid = ci >= 0;                         // wave-uniform
        const int chunk = c0 + ci;
        const bool padded = chunk >= a.first_pad_chunk;     // zero-padded rows score exactly 0
        if constexpr (FUSE) {
            // The bound of match_select_half_kernel -- best exact score of the chunk <= x + A + B_c + r_q R_c -- against the gate, as a
            // threshold on x (one more 1e-6 for the rearrangement; x is the fp32 score itself: the records round it up to their
            // 2^-20 grid, which MX6_SLACK pays for either way).  Each half-wave tests the maximum of ITS 64 rows and the two halves
            // meet as scalar masks: no cross-lane exchange, no per-lane branch (the ablations put the first form of this emit -- swap,
            // compare, exec-masked branch per query set -- at 10 % of the kernel).  m = queries of the set with a surviving chunk.
            // The thresholds are there already (next_thr, a tile earlier): what is left behind the fold is two compares, the scalar masks
            // and one branch, issued behind the first MFMAs of the tile this runs in.
            unsigned m[NS];
            unsigned many = 0u;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const unsigned long long hit = __ballot(!(s1[j] < thr[j]));
                m[j] = ((unsigned)hit | (unsigned)(hit >> 32)) & livemask[j];
                many |= m[j];
                s1[j] = -__builtin_inff();
            }
#ifdef VFM_ABL_NOCMP   // (timing experiment: nothing ever survives)
            many = 0u;
#endif
            if (valid && many != 0u) {   // rare: 0.56 survivors per query and 1564 chunks on D.2 data
                // (a list that has overflowed -- descriptors that are all alike -- takes no more entries: the search's guard goes up at
                // the end of the workgroup and match_gatepass_kernel decides every query; `seen` is a chunk old, the cap is exact)
#pragma unroll
                for (int j = 0; j < NS; ++j)
                    if (lane < 32 && ((m[j] >> lane) & 1u) && seen < (unsigned)LCAP) {
                        const unsigned pos = atomicAdd(&llist[0], 1u);
                        if (pos < (unsigned)LCAP) llist[1 + pos] = ((unsigned)ci << (NS == 2 ? 9 : 10)) | (unsigned)((wave * NS + j) * 32 + lane);
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if constexpr (FUSE) continue;
            unsigned best;
            if constexpr (TOP2) {
                // merge the two half-waves (coarse_emit_chunk's rule: the larger packed value, ties to the lower half), then to fixed point
                const int hi = lane >> 5;
                const unsigned o1 = swap32(t1[j]), o2 = swap32(t2[j]);
                const bool own = (t1[j] > o1) || (t1[j] == o1 && hi == 0);
                const unsigned w1 = own ? t1[j] : o1;
                const int wh = own ? hi : (1 - hi);
                const unsigned w2 = max(max(t2[j], o2), min(t1[j], o1));
                const int code = 63 - (int)(w1 & 63u);
                const int li = (code >> 4) * 32 + (code & 3) + 8 * ((code & 15) >> 2) + 4 * wh;  // row inside the chunk
                const int f1 = (int)ceilf((fmaxf(__uint_as_float(w1 | 63u), 0.0f) - 2.0f) * 1048576.0f) + I8_OFFSET;
                const int f2 = (int)ceilf((fmaxf(__uint_as_float(w2 | 63u), 0.0f) - 2.0f) * 1048576.0f) + I8_OFFSET;
                if (lane < 32 && qt0 + j < a.nq_tiles && valid)
                    a.partials[((size_t)(qt0 + j) * a.nchunks + (size_t)chunk) * 32 + lane] = make_uint2(((unsigned)f1 & ~127u) | (unsigned)li, (unsigned)f2);
                t1[j] = 0u;
                t2[j] = 0u;
                best = (unsigned)f1 & ~127u;
            } else {
                // the lane's best fp32 score of the chunk -> fixed point, rounded up; [query tile][chunk][32]
                const int fix = (int)ceilf(fmaxf(s1[j], -4.0f) * 1048576.0f);
                const unsigned rec = (unsigned)(fix + I8_OFFSET);
                best = umax_halves(rec);
                if (lane < 32 && qt0 + j < a.nq_tiles && valid)
                    reinterpret_cast<unsigned*>(a.partials)[((size_t)(qt0 + j) * a.nchunks + (size_t)chunk) * 32 + lane] = best;
                s1[j] = -__builtin_inff();
            }
            if (LOW && valid) {
                const int sbest = (int)best - I8_OFFSET;
                const float low = __builtin_fmaf(MX6_FIX_STEP * tab.x, (float)sbest, -(fx_A[j] + fx_mult[j] * tab.y));
                const float cand = (!padded || sbest > 0) ? low : -__builtin_inff();
                fx_arg[j] = cand > fx_low[j] ? chunk : fx_arg[j];   // (VFM_RECORDS_MX6_PILOT: the chunk the running lower bound comes from)
                fx_low[j] = fmaxf(fx_low[j], cand);
            }
        }
        const int cn = ci + 1 < nch ? ci + 1 : nch - 1;
#ifndef VFM_ABL_NOTAB
        tab = ltab[cn];   // for the next call (broadcast reads, consumed a chunk later: nothing waits for them here)
        if constexpr (FUSE) seen = llist[vzero];   // (through a per-lane zero: a uniform read would be moved to an SGPR on the spot -- lgkmcnt(0))
#else
        (void)cn;
#endif
    };

    floatx16 accA[NS], accB[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[j][r] = accB[j][r] = TOP2 ? 0.0f : -__builtin_inff();   // (folded by the first tile: no effect)

    // (A/B, MI355X_MICROARCH.md "static priority for the younger half": the second-dispatched half of an 8-wave workgroup loses the
    // VALU arbitration on every segment)
    if ((a.tune & 1) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    // everything the loop reads from registers is here before it starts: the compiler puts its own s_waitcnt vmcnt(0) in front of the
    // first use of a loaded value, and inside the loop that would wait for every piece in flight, step after step
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        asm volatile("" ::"v"(fx_A[j]), "v"(fx_mult[j]), "v"(fx_rq[j]), "v"(qs[j].x), "v"(qs[j].y), "v"((int)live[j]));
#pragma unroll
        for (int s = 0; s < KS6; ++s) asm volatile("" ::"v"(qf[j][s].c[0]), "v"(qf[j][s].c[1]), "v"(qf[j][s].c[2]), "v"(qf[j][s].c[3]), "v"(qf[j][s].c[4]), "v"(qf[j][s].c[5]));
    }
    // step 0 has landed (the barrier of step 0 waits for step 1, and so on): the other pre-loop steps may fly on
    if (ntiles > (RING - 2) * T) {
        if (my_pieces == PWMAX) wait_vmcnt_n<(RING - 2) * PWMAX>();
        else wait_vmcnt_n<(RING - 2) * (PWMAX - 1)>();
    } else {
        wait_vmcnt_n<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // this lane's slots of the code planes and of the scale plane of LDS tile 0 of ring slot 0
    const unsigned char* la0 = smem + MX6_SCALE_PLANE + 16 * lane;
    const unsigned char* lb0 = smem + MX6_SCALE_PLANE + 1024 + 8 * lane;
    const unsigned char* ls0 = smem + 8 * lane;
    Mx6Frag fr[PF];  // fragment ring registers of the tile in progress: k-step s consumes slot s % PF
    uint2 sc_cur, sc_nxt;   // block scales of the tile in progress / of the tile after it
#pragma unroll
    for (int s = 0; s < PF; ++s) fr[s] = mx6_frag(la0, lb0, s);
    sc_nxt = *reinterpret_cast<const uint2*>(ls0);
    tab = ltab[0];
    int ring = 0;   // ring slot (in steps) of the step in progress

    for (int it = 0; it < ntiles; it += T) {
        // (signed scalar arithmetic: the unsigned "== 0 ? RING - 1 : ring - 1" form became a VALU subtract-with-borrow + readfirstlane)
        int ring1 = ring + 1, ringL = ring - 1;      // ringL: slot of step k - 1 (k = it / T) = of step k + RING - 1
        if (ring1 >= RING) ring1 -= RING;
        if (ringL < 0) ringL += RING;
        int ringLL = ringL - 1;                      // slot of step k - 2 = of step k + RING - 2
        if (ringLL < 0) ringLL += RING;
        const unsigned cur_b = (unsigned)(ring * STEP_BYTES), nxt_b = (unsigned)(ring1 * STEP_BYTES);
        // The pieces of step k + RING - 1 go out in the WINDOW behind this step's barrier -- the k-step slots of its last tile, then
        // those of the next step's first tiles -- into the slot of step k - 1, which every wave has left when it passes that barrier.
        // Step j must have landed at the barrier of step j - 1 (the last tile's look-ahead reads its first fragments): RING - 2
        // whole steps after its window opened.
        const bool more = it + (RING - 1) * T < ntiles;                 // step k + RING - 1 exists: this step's window issues it
        const bool more_prev = it >= T && it + (RING - 2) * T < ntiles; // the window that opened in step k - 1 (it closes at this barrier) issues
        const bool lax = it + (RING - 2) * T < ntiles;                  // step k + RING - 2 was issued (by that window; the pre-loop, for k = 0)
        const int ck = it >> 2;   // first chunk of the step (of the unit)
        // tile J of the step into `acc`, folding `done` (the tile before it)
        auto tile = [&](auto Jc, auto Pc) __attribute__((always_inline)) {
            // (Pc: which accumulator set this tile fills -- 0: accA, folding accB; 1: the other way round.  Passed as arrays of NS
            // accumulators the lambda's parameter types made the HOST pass drop every instantiation of the kernel without a diagnostic.)
            constexpr int P = decltype(Pc)::value;
            floatx16 (&acc)[NS] = P ? accB : accA;
            const floatx16 (&done)[NS] = P ? accA : accB;
            constexpr int J = decltype(Jc)::value;
            const unsigned tb = cur_b + J * LT;
            const unsigned tn = (J + 1 < T) ? cur_b + (J + 1) * LT : nxt_b;  // the tile after it (stale after the last step)
            sc_cur = sc_nxt;
#ifndef VFM_ABL_NOLDS
            sc_nxt = *reinterpret_cast<const uint2*>(ls0 + tn);
#endif
            auto kstep = [&](auto Sc) __attribute__((always_inline)) {
                constexpr int s = decltype(Sc)::value;
                if constexpr (s == 0) {   // the tile's first MFMAs start from an inline constant (0, or 2.0): no accumulator to clear
                    floatx16 zero;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zero[r] = ACC0;
#pragma unroll
                    for (int j = 0; j < NS; ++j) acc[j] = mfma_mx6<s>(fr[s % PF], sc_cur, qf[j][s], qs[j], zero);
                } else {
#pragma unroll
                    for (int j = 0; j < NS; ++j) acc[j] = mfma_mx6<s>(fr[s % PF], sc_cur, qf[j][s], qs[j], acc[j]);
                }
                // (an empty use of the results: without it the MFMAs -- pure functions to the compiler -- are sunk across the
                // blocks of the step to their first reader, the fold one tile later)
#pragma unroll
                for (int j = 0; j < NS; ++j) asm volatile("" ::"v"(acc[j]));
#ifndef VFM_ABL_NOLDS
                fr[s % PF] = (s + PF < KS6) ? mx6_frag(la0 + tb, lb0 + tb, s + PF) : mx6_frag(la0 + tn, lb0 + tn, s + PF - KS6);
#else
                asm volatile("" : "+v"(fr[s % PF].c[0]));   // (the fragment "changes": the MFMAs are not loop-invariant to the compiler)
#endif
#ifndef VFM_ABL_NOEMIT
                // The chunk that ended with the previous tile's slots is emitted HERE -- behind the first two MFMAs of the second tile of
                // the next chunk, in front of this slot's folds (which start that chunk's maxima) --, not between two tiles: the matrix
                // pipe has 64 cycles of work while the compares, the scalar masks and the branch resolve.  (Between the tiles it was a
                // hole in every wave of the workgroup at the same time: 10 % of the kernel in tools/ablate6.py's timings.)
                // (No scheduling barriers around it: the compiler spreads the compares and the scalar masks through the slot's folds.  With the
                // two barriers -- rounds 4 and 5 -- the kernel is 1.5 % slower, tools/ab_two_libs_coarse.py; VFM_ABL_EMITSB puts them back.)
                if constexpr (s == 0 && (J & 3) == 1) {
#ifdef VFM_ABL_EMITSB
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    emit_chunk(ck + (J >> 2) - 1);
#ifdef VFM_ABL_EMITSB
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
#endif
                if constexpr (s == 0 && (J & 3) == 3) next_thr();
#ifndef VFM_ABL_NOFOLD
#pragma unroll
                for (int e = s * (16 * NS) / KS6; e < (s + 1) * (16 * NS) / KS6; ++e) {   // `done` is tile (J + 3) & 3 of its chunk
                    if constexpr (TOP2) coarse_fold(t1[e >> 4], t2[e >> 4], done[e >> 4][e & 15], ((J + 3) & 3) * 16 + (e & 15));
                    else s1[e >> 4] = fmaxf(s1[e >> 4], done[e >> 4][e & 15]);
                }
#else
                if (s == 0) {
#pragma unroll
                    for (int j = 0; j < NS; ++j) s1[j] = fmaxf(s1[j], done[j][0]);
                }
#endif
                // (the same for the fold: the running maxima are read at the end of the chunk only, and the compiler had moved the
                // folds of tiles 1 - 3 there -- one block of 45 dependent v_max3 behind the barrier, in every wave at the same time)
                // (uses, not definitions: the compiler keeps knowing what the values are -- no re-canonicalisation of the floats)
                if constexpr (TOP2) asm volatile("" ::"v"(t1[0]), "v"(t1[1]), "v"(t2[0]), "v"(t2[1]));
                else {
#pragma unroll
                    for (int j = 0; j < NS; ++j) asm volatile("" ::"v"(s1[j]));
                }
                // one 1 KiB piece per k-step slot of the window: the last tile of the step carries slots 0 .. KS6 - 1, tile J of the
                // next step slots KS6 (J + 1) ..
                constexpr int kslot = (J == T - 1 ? 0 : KS6 * (J + 1)) + s;
                if constexpr (kslot < PWMAX) {
                    __builtin_amdgcn_sched_barrier(0);
#ifndef VFM_ABL_NODMA
                    if (kslot < my_pieces && (J == T - 1 ? more : more_prev)) issue_piece(kslot, (unsigned)((J == T - 1 ? ringL : ringLL) * STEP_BYTES));
#endif
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
            // (both registers of the next tile's scales stay allocated until here: with KS6 <= 4 the second one is never read, the
            // register allocator handed it out again right behind the ds_read_b64, and the wave waited on lgkmcnt(0) to overwrite it)
            asm volatile("" ::"v"(sc_nxt.x), "v"(sc_nxt.y));
        };
        tile(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});   // (its slots fold the last tile of the previous chunk; tile 1 emits that chunk)
        tile(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        tile(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
        if constexpr (T == 8) {
            tile(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
            tile(std::integral_constant<int, 4>{}, std::integral_constant<int, 0>{});
            tile(std::integral_constant<int, 5>{}, std::integral_constant<int, 1>{});
            tile(std::integral_constant<int, 6>{}, std::integral_constant<int, 0>{});
        }
        if (more_prev) gstep += T * IMG_TB;   // the previous window is closed: from here on gstep is the step THIS step's window issues
#ifndef VFM_ABL_NOBAR
        // the next step's pieces have landed: at most the pieces of the RING - 3 steps issued since (this wave's own count each)
        // are still in flight; where the unit ends and steps stop being issued, everything is waited for
        if (lax && RING > 3) {
            if (my_pieces == PWMAX) wait_vmcnt_n<(RING - 3) * PWMAX>();
            else wait_vmcnt_n<(RING - 3) * (PWMAX - 1)>();
        } else {
            wait_vmcnt_n<0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#endif
        tile(std::integral_constant<int, T - 1>{}, std::integral_constant<int, 1>{});
        ring = ring1;
    }
#pragma unroll
    for (int e = 0; e < 16 * NS; ++e) {
        if constexpr (TOP2) coarse_fold(t1[e >> 4], t2[e >> 4], accB[e >> 4][e & 15], 3 * 16 + (e & 15));
        else s1[e >> 4] = fmaxf(s1[e >> 4], accB[e >> 4][e & 15]);
    }
    next_thr();
    emit_chunk(nch - 1);
    if constexpr (LOW) {
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (lane < 32 && qt0 + j < a.nq_tiles) {
                atomicMax(a.qmax + (size_t)(qt0 + j) * 32 + lane, float_key(fx_low[j]));
                if (a.qbest && fx_low[j] > -__builtin_inff())
                    atomicMax(a.qbest + (size_t)(qt0 + j) * 32 + lane, ((unsigned long long)float_key(fx_low[j]) << 32) | (unsigned)fx_arg[j]);
            }
    }
    if constexpr (FUSE) {
        // the workgroup's list -> its slot of the survivor buffer: [count, query block, first chunk, -][entries]; plain stores
        __syncthreads();
        const unsigned total = llist[0];
        const unsigned cnt = total < (unsigned)LCAP ? total : (unsigned)LCAP;
        unsigned* slot = a.surv + (size_t)blockIdx.x * (MX6_LCAP + 1 + 3);
        if (threadIdx.x == 0) {
            if (blockIdx.x == 0) a.survivors[MX6_GRID_SLOT - 5] = (int)gridDim.x;   // fb_count[MX6_GRID_SLOT]: how many slots there are
            slot[0] = cnt;
            slot[1] = (unsigned)qb;
            slot[2] = (unsigned)c0;
            slot[3] = (total >= (unsigned)LCAP ? 1u : 0u) | ((unsigned)NS << 8);   // (bits 8 ..: query tiles per wave -- how match_bin_survivors_kernel reads the entries)
            if (total >= (unsigned)LCAP) a.survivors[HALF_GUARD_FLAG - 5] = 1;   // fb_count[HALF_GUARD_FLAG]
        }
        for (unsigned i = threadIdx.x; i < cnt; i += 512) slot[4 + i] = llist[1 + i];
    }
#endif   // __HIP_DEVICE_COMPILE__
}

template <int KS6, int KIND, bool LOW, int IMG_KS6, int RING, int T = 4, int NS = 2>
int launch_mx6q2(const CoarseArgs& a, hipStream_t st) {
    constexpr int LT = MX6_SCALE_PLANE + KS6 * MX6_KSTEP_BYTES;
    constexpr int LCAP = RING * T * LT + MX6_LTAB * 8 + (MX6_LCAP + 1) * 4 <= 160 * 1024 ? MX6_LCAP : 1023;
    const int lds = RING * T * LT + MX6_LTAB * 8 + (KIND == MX6_FUSE ? (LCAP + 1) * 4 : 0);
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_coarse_mx6q2_kernel<KS6, KIND, LOW, IMG_KS6, RING, T, NS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_mark(attr_set);
    }
    hipLaunchKernelGGL((match_coarse_mx6q2_kernel<KS6, KIND, LOW, IMG_KS6, RING, T, NS>), dim3(a.nqb * a.nslices), dim3(512), lds, st, a);
    return VFM_OK;
}

}  // namespace

static_assert(MX6_LCAP + 1 + 3 == MX6_SURV_SLOT_WORDS, "carve_search sizes the slots");
int mx6_survivor_slot_words() { return MX6_SURV_SLOT_WORDS; }

// the fp6 coarse kernel for the arguments do_search_coarse prepared (a.Qh / a.Bh = the fp6 tiles, a.ib = mx6_bounds);
// d = 256 / 384 (full width) or 256 / 384 / 512 / 768 (half width) and more than 2048 queries (effective_records)
int launch_coarse_mx6(CoarseArgs& a, int d, bool top2, bool half, bool fuse, hipStream_t st) {
    const bool ns3 = half && fuse && d == 384 && vfm_cfg().mx6_ns3 && vfm_cfg().mx6_t4;
    a.nqb = ns3 ? (a.nq_tiles + 23) / 24 : (a.nq_tiles + 15) / 16;
    a.nslices = choose_slices(a.nqb, a.nchunks);
    if (half && vfm_cfg().force_slices == 0) {
        // the half-width kernel's workgroups are short, and shorter ones let the other stages of a pipeline in: ~7.5 rounds of 256
        // workgroups instead of choose_slices' 5 (tools/sweep_slices.py, C2, 200 steps: 48 slices 1548 against 32 slices 1475
        // registrations/s; the full-width fp6 kernel and the int8 kernels are flat from 32 on)
        int s = (int)((1600 + a.nqb - 1) / a.nqb);   // (round 4, tools/sweep_slices_r4.py, fused kernel: 40 / 44 slices 1764 / 1762, 48: 1740, 57: 1690 registrations/s over 200 steps)
        const int smax = a.nchunks / 8 < 64 ? a.nchunks / 8 : 64;
        s = s > smax ? smax : s;
        s = s < 1 ? 1 : s;
        // Round 6: the workgroups are equal and a compute unit holds one, so the kernel lasts ceil(workgroups / units) rounds whatever the last
        // round holds (tools/sweep_slices_r6.py, the kernel alone at C2, 27 query blocks: 60 slices = 6.33 rounds 0.388 ms; 56 = 5.9 rounds
        // 0.370; 47 = 4.96 rounds 0.367; 19 slices = 2.004 rounds 0.445).  Near the count above, take the slice count whose last round is
        // (almost) full -- at most 4 above, 12 below; in the pipeline 56 against 60 is +1 % over 200 steps, even in the 20-step form.
        {
            static std::atomic<int> ncu{0};
            int cus = ncu.load(std::memory_order_relaxed);
            if (cus == 0) {
                int dev = 0;
                (void)hipGetDevice(&dev);
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
                ncu.store(cus, std::memory_order_relaxed);
            }
            int best = s;
            double best_fill = 0.0;
            for (int t = (s + 4 < smax ? s + 4 : smax); t >= 1 && t >= s - 12; --t) {
                const long long wg = (long long)a.nqb * t, rounds = (wg + cus - 1) / cus;
                const double fill = (double)wg / (double)(rounds * cus);
                if (fill >= 0.975) {   // the largest such count (short workgroups let the pipeline's other stages in)
                    best = t;
                    best_fill = 2.0;
                    break;
                }
                if (fill > best_fill) {
                    best_fill = fill;
                    best = t;
                }
            }
            s = best;
        }
        a.nslices = s < 1 ? 1 : s;
    }
    while ((a.nchunks + a.nslices - 1) / a.nslices + 1 > MX6_LTAB && a.nslices < a.nchunks) ++a.nslices;   // a slice's constants fit the LDS table
    const int slices_cap = a.nchunks / 8 < 1 ? 1 : (a.nchunks / 8 > 64 ? 64 : a.nchunks / 8), slices_tab = (a.nchunks + 254) / 255;
    if (fuse && (a.nqb > a.npad / 512 + 1 || a.nslices > (slices_cap > slices_tab ? slices_cap : slices_tab)))   // (what carve_search sized the slots for)
        return vfm_fail(VFM_EINVAL, "search_coarse: %d x %d survivor slots exceed the workspace (n %d chunks %d)", a.nqb, a.nslices, a.npad, a.nchunks);
    if (g_prof_start) VFM_CHECK_HIP(hipEventRecord(g_prof_start, st));
    int rc;
    // Two chunks per barrier (T = 8, ring of 3 x 8 tiles: 84 / 120 KiB at d = 256 / 384) is built, parity-tested and measured: 0.415 ms
    // against 0.419 alone, 1715 against 1725 registrations/s in the pipeline (tools/ab_r4.py) -- the barrier share the ablations
    // show (tools/ablate6.py: -10 % without barrier and staging) does not come back by halving the barriers, and the larger ring
    // leaves the side kernels less LDS.  One chunk per barrier stays the default; vfm_debug_set_coarse_variant(31) selects T = 8.
    const bool t8 = half && fuse && (d == 384 || d == 256) && !vfm_cfg().mx6_t4 && a.nslices <= a.nchunks / 2;
    a.tune = vfm_cfg().mx6_tune;
    if (ns3 && (vfm_cfg().mx6_tune & 2))   // (A/B) a ring of five steps
        rc = launch_mx6q2<3, MX6_FUSE, false, 6, 5, 4, 3>(a, st);
    else if (ns3)   // three query tiles per wave: 768 queries per workgroup (nqb set above)
        rc = launch_mx6q2<3, MX6_FUSE, false, 6, 4, 4, 3>(a, st);
    else if (fuse && !half)   // VFM_RECORDS_MX6_FUSED: the full-width pass with the gate test in its epilogue
        rc = d == 384 ? launch_mx6q2<6, MX6_FUSE, false, 6, 4>(a, st) : launch_mx6q2<4, MX6_FUSE, false, 4, 4>(a, st);
    else if (half && fuse && t8)
        rc = d == 384 ? launch_mx6q2<3, MX6_FUSE, false, 6, 3, 8>(a, st) : launch_mx6q2<2, MX6_FUSE, false, 4, 3, 8>(a, st);
    else if (half && fuse)
        rc = d == 768 ? launch_mx6q2<6, MX6_FUSE, false, 12, 4>(a, st) : d == 512 ? launch_mx6q2<4, MX6_FUSE, false, 8, 4>(a, st)
           : d == 384 ? launch_mx6q2<3, MX6_FUSE, false, 6, 4>(a, st) : launch_mx6q2<2, MX6_FUSE, false, 4, 4>(a, st);
    else if (half)
        rc = d == 768 ? launch_mx6q2<6, MX6_BEST, false, 12, 4>(a, st) : d == 512 ? launch_mx6q2<4, MX6_BEST, false, 8, 4>(a, st)
           : d == 384 ? launch_mx6q2<3, MX6_BEST, false, 6, 4>(a, st) : launch_mx6q2<2, MX6_BEST, false, 4, 4>(a, st);
    else
        rc = d == 384 ? (top2 ? launch_mx6q2<6, MX6_TOP2, true, 6, 4>(a, st) : launch_mx6q2<6, MX6_BEST, true, 6, 4>(a, st))
                      : (top2 ? launch_mx6q2<4, MX6_TOP2, true, 4, 4>(a, st) : launch_mx6q2<4, MX6_BEST, true, 4, 4>(a, st));
    if (rc) return rc;
    VFM_CHECK_LAUNCH("match_coarse_mx6q2_kernel");
    if (g_prof_stop) VFM_CHECK_HIP(hipEventRecord(g_prof_stop, st));
    g_prof_start = g_prof_stop = nullptr;
    return VFM_OK;
}

}  // namespace vfmm
