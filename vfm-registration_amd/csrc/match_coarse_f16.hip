// match_coarse_f16.hip -- the fp16 MFMA coarse pass of the matcher: queries resident in VGPRs, map
// tiles streamed through an LDS ring by LDS-DMA, sparse row-level records or per-chunk top-2 records.  Three shapes:
//   match_coarse_pipe_kernel  d <= 384: 8 waves, fragment pipeline carried across the step barrier
//   match_coarse_kernel       d = 512 (ring of 4), and the A/B + ablation base
//   match_coarse_r_kernel     d = 640 / 768: 4 waves, 192 query registers
#include "match_internal.h"

namespace vfmm {
namespace {

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


}  // namespace

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
        rc = (vfm_cfg().coarse_qsets == 2)   ? launch_coarse_v<KSTEPS, 2>(a, st)
             : (vfm_cfg().coarse_qsets == 1) ? launch_coarse_v<KSTEPS, 1>(a, st)
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

// the fp16 coarse kernel for the arguments do_search_coarse prepared (d in {128, ..., 768})
int launch_coarse_f16(const CoarseArgs& a, int d, hipStream_t st) {
    switch (d / 16) {
        case 8: return launch_coarse<8>(a, st);
        case 16: return launch_coarse<16>(a, st);
        case 24: return launch_coarse<24>(a, st);
        case 32: return launch_coarse<32>(a, st);  // 8-wave kernel, ring of 4: 3.45 ms at C2 x 512 (4-wave kernel: 4.42 ms)
        case 40: return a.row_bias ? launch_coarse_r<40, 1, 3, true>(a, st) : launch_coarse_r<40, 1, 3, false>(a, st);
        case 48: return a.row_bias ? launch_coarse_r<48, 1, 3, true>(a, st) : launch_coarse_r<48, 1, 3, false>(a, st);
        default: return vfm_fail(VFM_EINVAL, "FAST matching supports d in {128,256,384,512,640,768}, got %d", d);
    }
}

}  // namespace vfmm
