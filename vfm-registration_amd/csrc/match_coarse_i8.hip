// match_coarse_i8.hip -- the int8 MFMA coarse pass of the matcher: v_mfma_i32_32x32x32_i8 over the int8
// image of prep_chunk_kernel, exact integer scores, one record per (query, 128-row chunk) -- the best score, or the packed
// top-2 with the best row's index -- and per query a lower bound of its exact maximum.
//   match_coarse_i8q2_kernel  d = 256 / 384, more than 2048 queries: 64 resident queries per wave
//   match_coarse_i8_kernel    every other size and width: 32 resident queries per wave, T tiles per step
#include "match_internal.h"

namespace vfmm {
namespace {

// ---------------------------------------------------------------------------------------------
// int8 coarse pass with T map tiles per step (the schedule of match_coarse_pipe_kernel<., false, true>, generalised).
// One barrier and one staging round per T * KSTEPS MFMAs: at d = 768 (KSTEPS = 24, T = 2: 48 MFMAs per step) the int8
// kernel measured 0.53 of the int8 peak, at d = 384 (KSTEPS = 12, T = 2: 24 per step) 0.49 -- the per-step cost is fixed,
// so d = 384 / 256 take T = 4 here (one whole 128-row chunk per step, 48 / 32 MFMAs).  Ring = 3 steps of T tiles
// (in use | landed | in flight), fragment look-ahead PF = 2 k-steps for T = 4 (4 tiles x 2 x 4 registers), 4 for T = 2.
// ---------------------------------------------------------------------------------------------
// TOP2: the packed per-chunk top-2 records of the fp16 pass (best row index included, 3 VALU ops per element) instead of the
// best value alone: candidate chunks with one row inside the bounds need no rescan -- the choice for duplicate-rich maps.
// LOW = false (half-width pass): no running lower bound of the query's exact maximum (see match_coarse_i8q2_kernel)
template <int KSTEPS, int T, bool TOP2 = false, bool LOW = true>
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
        if (LOW && chunk >= 0) {  // wave-uniform
            // a chunk with zero-padded rows (they score exactly 0) counts only where its best score is positive: that score
            // belongs to a real row
            const float sb = a.ib.bstep[chunk], be = a.ib.berr[chunk];
            const int sbest = (int)best - I8_OFFSET;
            const float low = __builtin_fmaf(i8_sq * sb, (float)sbest, -(i8_A + i8_mult * be));
            i8_low = fmaxf(i8_low, (chunk < a.first_pad_chunk || sbest > 0) ? low : -__builtin_inff());
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
    if (LOW && lane < 32 && qt < a.nq_tiles) atomicMax(a.qmax + (size_t)qt * 32 + lane, float_key(i8_low));
}

// ---------------------------------------------------------------------------------------------
// int8 coarse pass, 64 resident queries per wave (d = 256 / 384: 2 x 48 query registers fit beside everything else).
// What the fp16 kernel could not afford (2 x 96 query registers) and its ablations named as the remaining cost: every map
// fragment read from LDS feeds TWO MFMAs and a workgroup covers 512 queries, so LDS operand reads and L2 -> LDS staging per
// MFMA both halve.  The four tiles of a step are processed one after the other (a tile = KSTEPS slots of two MFMAs, one per
// query set, sharing the fragment); the accumulators of a finished tile are folded in the slots of the next one, and two
// accumulator pairs alternate, so nothing is copied.  One barrier per 8 * KSTEPS MFMAs.
// ---------------------------------------------------------------------------------------------
// LOW = false (half-width pass): the running lower bound of the query's exact maximum is not kept -- its selection tests
// against the gate alone
// FUSE (half-width pass, VFM_RECORDS_HALF_FUSED): no records -- the chunk's best score is tested against the gate as soon as
// it exists (the bound of match_select_half_kernel) and a survivor goes straight into the chunk's rescan bin
template <int KSTEPS, bool TOP2 = false, bool LOW = true, bool FUSE = false>
__global__ __launch_bounds__(512, 2) void match_coarse_i8q2_kernel(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVES = 8, T = 4;
    constexpr int TILE_U4 = KSTEPS * 64;
    constexpr int TILE_BYTES = TILE_U4 * 16;
    constexpr int PIECES = T * KSTEPS / NWAVES;  // 1 KiB pieces per wave per step: 6 (d = 384) or 4 (d = 256)
    constexpr int NBUF = 3 * T;
    constexpr int PF = (KSTEPS % 4 == 0 && KSTEPS >= 8) ? 4 : (KSTEPS % 3 == 0 ? 3 : 2);  // fragment look-ahead in k-steps
    static_assert(KSTEPS % PF == 0 && KSTEPS >= 2 * PF && PIECES <= 2 * T && (T * KSTEPS) % NWAVES == 0, "shape");
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
    float fuse_rq[2] = {0.f, 0.f};
    bool fuse_live[2] = {false, false};
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
        if constexpr (FUSE) {
            fuse_rq[j] = a.qrest[qi];
            fuse_live[j] = qt0 + j < a.nq_tiles && (int64_t)qi < a.n_valid && a.qinv[qi] != 0.0f;
        }
    }
    stage_step(gsrc, 0u);
    if (ntiles > T) stage_step(gsrc + (size_t)T * TILE_U4, (unsigned)(T * TILE_BYTES));
    const uint4* gnext = gsrc + (size_t)2 * T * TILE_U4;

    unsigned s1[2] = {0u, 0u}, s2[2] = {0u, 0u}, unused_max = 0u;
    auto emit_chunk = [&](int chunk) __attribute__((always_inline)) {  // chunk < 0: nothing folded yet
        float sb = 0.f, be = 0.f, rb = 0.f;
        const bool counted = (LOW || FUSE) && chunk >= 0;  // wave-uniform
        if (counted) {
            sb = a.ib.bstep[chunk];
            be = a.ib.berr[chunk];
            if constexpr (FUSE) rb = a.grest[chunk];
        }
        // a chunk with zero-padded rows (they score exactly 0) counts only where its best score is positive: that score
        // belongs to a real row
        const bool padded = chunk >= a.first_pad_chunk;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (FUSE) {
                const unsigned w1 = max(s1[j], (unsigned)__shfl_xor(s1[j], 32));   // the chunk's best score (all lanes)
                s1[j] = 0u;
                if (counted && lane < 32 && fuse_live[j]) {
                    // the bound of match_select_half_kernel: best exact score of the chunk <= s_q s_c S_half + A + B_c + r_q R_c
                    const float up = (i8_sq[j] * sb) * (float)((int)w1 - I8_OFFSET) + (i8_A[j] + i8_mult[j] * be) + (fuse_rq[j] * rb + 1.0e-6f);
                    if (!(up < a.gate)) {   // rare: a handful per query where the pass is used at all
                        const int64_t q = (int64_t)(qt0 + j) * 32 + lane;
                        // (the search's load figure is summed from the bin counts by match_rescan_chunk_kernel: ten thousand
                        // same-address atomics from inside this kernel cost it more than the selection kernel it replaces)
                        // (a chunk that has collected FUSE_BIN_SATURATE_X times its bin capacity already -- descriptors that are all alike -- stops
                        // recording: the entry is dropped and the search's guard flag raised, match_gatepass_kernel then decides
                        // every query; without the cut-off such data cost this kernel 31 million atomics)
                        const unsigned seen = __hip_atomic_load(&a.bin_cnt[(size_t)chunk * BIN_CNT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned saturate = (unsigned)(FUSE_BIN_SATURATE_X * a.bin_cap);
                        const unsigned pos = seen >= saturate ? seen : atomicAdd(&a.bin_cnt[(size_t)chunk * BIN_CNT_STRIDE], 1u);
                        if (pos < (unsigned)a.bin_cap) {
                            a.bins[(size_t)chunk * a.bin_cap + pos] = (int)q;
                        } else if (pos < saturate) {   // a full bin leaves the entry in the query's own list (match_rescan_kernel)
                            const int slot = atomicAdd(&a.cand_cnt[q], 1);
                            if (slot < a.cap) a.cand[(size_t)q * a.cap + slot] = ((unsigned)chunk << 8) | 128u;
                        } else {
                            a.survivors[HALF_GUARD_FLAG - 5] = 1;   // fb_count[HALF_GUARD_FLAG] (a.survivors = fb_count + 5)
                        }
                    }
                }
                continue;
            }
            const unsigned best = TOP2 ? coarse_emit_chunk(a, s1[j], s2[j], unused_max, qt0 + j, chunk)
                                       : coarse_emit_chunk_best(a, s1[j], qt0 + j, chunk);
            if (counted) {
                const int sbest = (int)best - I8_OFFSET;
                const float low = __builtin_fmaf(i8_sq[j] * sb, (float)sbest, -(i8_A[j] + i8_mult[j] * be));
                i8_low[j] = fmaxf(i8_low[j], (!padded || sbest > 0) ? low : -__builtin_inff());
            }
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
    if constexpr (LOW) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (lane < 32 && qt0 + j < a.nq_tiles) atomicMax(a.qmax + (size_t)(qt0 + j) * 32 + lane, float_key(i8_low[j]));
    }
}


}  // namespace

template <int KSTEPS, int T, bool TOP2 = false, bool LOW = true>
int launch_coarse_i8(const CoarseArgs& a, hipStream_t st) {
    const int lds = 3 * T * KSTEPS * 1024;
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_coarse_i8_kernel<KSTEPS, T, TOP2, LOW>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_mark(attr_set);
    }
    hipLaunchKernelGGL((match_coarse_i8_kernel<KSTEPS, T, TOP2, LOW>), dim3(a.nqb * a.nslices), dim3(512), lds, st, a);
    return VFM_OK;
}

template <int KSTEPS, bool TOP2, bool LOW = true, bool FUSE = false>
int launch_coarse_i8q2(const CoarseArgs& a, hipStream_t st) {
    const int lds = 12 * KSTEPS * 1024;
    static unsigned long long attr_set = 0ull;  // one bit per device
    if (!attr_done(attr_set)) {
        VFM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&match_coarse_i8q2_kernel<KSTEPS, TOP2, LOW, FUSE>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_mark(attr_set);
    }
    hipLaunchKernelGGL((match_coarse_i8q2_kernel<KSTEPS, TOP2, LOW, FUSE>), dim3(a.nqb * a.nslices), dim3(512), lds, st, a);
    return VFM_OK;
}


// the int8 coarse kernel for the arguments do_search_coarse prepared; sets the query-block / slice split of the shape it picks
int launch_coarse_int8(CoarseArgs& a, int d, int64_t n, int records, hipStream_t st) {
    a.nqb = (int)(rows_padded(n) / QBLOCK);  // 8 waves x 32 queries at every width
    a.nslices = choose_slices(a.nqb, a.nchunks);
    if (g_prof_start) VFM_CHECK_HIP(hipEventRecord(g_prof_start, st));
    const bool top2 = records == VFM_RECORDS_TOP2;
    int rc8;
    if (records == VFM_RECORDS_HALF_FUSED) {   // d = 256 / 384, more than 2048 queries (effective_records)
        a.nqb = (a.nq_tiles + 15) / 16;
        a.nslices = choose_slices(a.nqb, a.nchunks);
        rc8 = d == 384 ? launch_coarse_i8q2<6, false, false, true>(a, st) : launch_coarse_i8q2<4, false, false, true>(a, st);
    } else if (records == VFM_RECORDS_HALF) {
        // the half-width pass: the same kernels on the image of the first d / 2 columns (a.Qh / a.Bh = tiles8h), best-score records
        if (n > 2048 && vfm_cfg().coarse_qsets == 0) {   // 64 resident queries per wave at every width: d / 2 columns are at most 384
            a.nqb = (a.nq_tiles + 15) / 16;
            a.nslices = choose_slices(a.nqb, a.nchunks);
            switch (d / 64) {
                case 4: rc8 = launch_coarse_i8q2<4, false, false>(a, st); break;
                case 6: rc8 = launch_coarse_i8q2<6, false, false>(a, st); break;
                case 8: rc8 = launch_coarse_i8q2<8, false, false>(a, st); break;
                case 10: rc8 = launch_coarse_i8q2<10, false, false>(a, st); break;
                default: rc8 = launch_coarse_i8q2<12, false, false>(a, st); break;
            }
        } else {   // one query set per wave: few queries (variants 10 / 12: every size, A/B)
            switch (d / 64) {
                case 4: rc8 = launch_coarse_i8<4, 4, false, false>(a, st); break;
                case 6: rc8 = launch_coarse_i8<6, 4, false, false>(a, st); break;
                case 8: rc8 = launch_coarse_i8<8, 4, false, false>(a, st); break;
                case 10: rc8 = launch_coarse_i8<10, 4, false, false>(a, st); break;
                default: rc8 = launch_coarse_i8<12, 4, false, false>(a, st); break;
            }
        }
    } else if (d <= 384 && n > 2048 && vfm_cfg().coarse_qsets == 0) {
        // 64 resident queries per wave: 11-15 % faster than the one-set kernel from ~3000 queries on (C2: 1.09 vs 1.23 ms;
        // 1500 x 100 000: 0.059 vs 0.056 ms -- half as many, twice as large workgroups); variants 10 / 12 = one set, A/B
        a.nqb = (a.nq_tiles + 15) / 16;
        a.nslices = choose_slices(a.nqb, a.nchunks);
        rc8 = d == 384 ? (top2 ? launch_coarse_i8q2<12, true>(a, st) : launch_coarse_i8q2<12, false>(a, st))
                       : (top2 ? launch_coarse_i8q2<8, true>(a, st) : launch_coarse_i8q2<8, false>(a, st));
    } else {
        const bool t2 = vfm_cfg().coarse_qsets == 10;  // variant 10 (A/B): 2 tiles per step at every width
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

}  // namespace vfmm
