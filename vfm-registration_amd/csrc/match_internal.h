// match_internal.h -- declarations shared by the translation units of the matcher (csrc/match_*.hip): constants, the
// argument block and device helpers of the coarse kernels, the layout of prepared operands and search workspaces, and the
// host functions that cross translation units.  Kernels live in anonymous namespaces of the file that launches them.
#pragma once
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

namespace vfmm {

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
// candidate queries a map chunk can collect for the chunk-major int8 rescan (the rest: query-major): sized per search from the
// queries per chunk -- 128 per average query share, at least 1024 (C2: 1664; the reverse direction of a Euclidean search, 200 000
// queries on 157 chunks: 65 536; round 2's fixed 512 sent such searches through the 48-KB-per-candidate query-major path)
__host__ __device__ inline int rescan_bin_cap(int64_t npad, int64_t nchunks) {
    int64_t c = (128 * npad / (nchunks > 0 ? nchunks : 1) + 63) / 64 * 64;
    return (int)(c < 1024 ? 1024 : (c > 65536 ? 65536 : c));
}
// One counter per 128-byte line: device-scope atomics are resolved at the memory side, line by line -- 1563 counters in 49
// consecutive lines took ~0.3 ns per atomic whatever the number of threads (select_best: 64 us of 122 at 250 000 candidates, 283 of
// 345 at 920 000).
constexpr int BIN_CNT_STRIDE = 32;
constexpr int RESCAN_SLICE = 256;    // bin entries one workgroup of match_rescan_chunk_kernel takes (grid.y slices a long bin; 1024 until the end of round 4: the longest workgroups -- 32 blocks of 2.6 us -- were the kernel, 155 -> 140 us at 590 queries per chunk; 128 pays more prologues than it balances)
constexpr int RESCAN_BATCH = 64;     // ... of which match_rescan_chunk_kernel stages this many in LDS at a time
// Half-width pass, device-side guard: a search whose bound leaves more than this many (query, chunk) pairs per query -- descriptors
// that are all alike: every chunk survives -- does not rescan them (31 million 128-row rescans at C2: 171 ms) but falls through,
// inside the same _finish call, to match_gatepass_kernel: one full-width int8 MFMA pass with the gate as hit test (2-3 ms).
// SearchWs::fb_count[7] is the flag (half_guard_kernel); the host policy (vfmreg/pipeline.py) leaves the mode on the same figure.
constexpr int HALF_GUARD_PER_QUERY = 48;
constexpr int HALF_GUARD_FLAG = 7;   // index into fb_count
constexpr int MX6_SURV_SLOT_WORDS = 2047 + 4;   // fused fp6 half-width pass: a workgroup's survivor slot = header (4 words) + up to 2047 entries
constexpr int MX6_GRID_SLOT = 32;    // fb_count[.]: workgroups of the fused fp6 half-width kernel = survivor slots match_bin_survivors_kernel walks
constexpr int FUSE_BIN_SATURATE_X = 8;  // fused form: a chunk that collected this many times its bin capacity stops recording (and raises the flag)
constexpr int FILTER_LDS_ROWS = 1024;  // sparse fp16 records a query can hold (= SearchWs::rcap; match_filter_refine_kernel keeps them in LDS)
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

// The same, 4 B per lane: LDS destination = M0 + lane * 4 (a gather of one dword per lane).
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off\n\t"
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
// Euclidean mode of the int8 finish stage (row A6, match_l2i8.hip; qn == NULL: inner-product mode).  The int8 image holds the
// NORMALISED rows, so the proven bound  cosU = s_q s_c S + A + B_c  bounds the cosine; with the commonly scaled norms |a~|, |b~|
// (<= 1) the Euclidean score  a~.b~ - |b~|^2 / 2  of a row is at most  qn * bn * max(cosU, 0) - bn^2 / 2 + slack  (slack covers
// the fp32 roundings of the norms: (d + 16) 2^-22).  Map rows are SORTED by norm, so a chunk's norms span a narrow [lo, hi].
struct L2Terms {
    const float* qn;   // [npad] |a~| per query
    const float* bn;   // [mpad] |b~| per map row, in the (sorted) row order of the int8 image
    float slack;
};
__device__ __forceinline__ float l2_upper_row(float qn, float bn, float cosU, float slack) {
    return __builtin_fmaf(qn * bn, fmaxf(cosU, 0.0f), -0.5f * bn * bn) + slack;
}

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
    // half-width pass with the selection fused into the coarse kernel (VFM_RECORDS_HALF_FUSED): no records are written --
    // a (query, chunk) pair whose bound reaches `gate` goes straight into the chunk's bin (or, past the bin, the query's list)
    float gate;
    int64_t n_valid;        // real queries
    const float* qrest;     // [npad] |second half of the normalised query|, rounded up
    const float* grest;     // [nchunks] its maximum over the chunk's rows
    unsigned* bin_cnt;      // [nchunks * BIN_CNT_STRIDE], zeroed per search
    int* bins;              // [nchunks][bin_cap]
    int bin_cap;
    int* cand_cnt;          // [npad], zeroed per search (entries of the query's own list)
    unsigned* cand;         // [npad][cap]
    int cap;
    int* survivors;         // fb_count + 5: the search's load figure
    unsigned long long* qbest;   // VFM_RECORDS_MX6_PILOT: [npad] (float_key(lower bound) << 32 | chunk) of the query's best chunk, by 64-bit atomicMax (NULL: not kept)
    int tune;               // (A/B, vfm_config "mx6_tune") bit 0: waves 4 - 7 of a workgroup of the fp6 kernel at s_setprio 1
    unsigned* surv;         // VFM_RECORDS_MX6_HALF_FUSED / _MX6_FUSED: a slot of mx6_survivor_slot_words() words per workgroup (in the record buffer)
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
        // int8 pass: [query tile][chunk][32] (a query tile's records are one contiguous stream for match_select_top2_kernel);
        // fp16 pass: [chunk][npad]
        const size_t at = a.ib.qerr ? ((size_t)qt * a.nchunks + (size_t)chunk) * 32 + lane : (size_t)chunk * a.npad + (size_t)qt * 32 + lane;
        a.partials[at] = make_uint2((w1 & ~127u) | (unsigned)li, w2);
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

// arg-max over a wavefront, ties -> lowest index (the oracle's rule)
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

// =============================================================================================
// host side
// =============================================================================================
struct Prepared {
    float* inv;
    uint4* tiles;
    // int8 image (i8_capable(d); prep_chunk_kernel)
    float* err;       // E per row
    float* gstep;     // quantisation step per group of 128 rows
    float* gerr;      // maximum E per group
    uint4* tiles8;    // int8 fragment tiles
    uint4* tiles8h;   // int8 fragment tiles of the first d / 2 columns
    float* rest;      // per row: |second half of the normalised row|_2, rounded up
    float* grest;     // its maximum per group
    // fp6 image (mx6_width(d); prep_chunk_kernel<., ., true>, written on request: VFM_PREPARE_MX6)
    uint4* tiles6;    // MX fp6 (e2m3) fragment tiles of the 32x32x64 scaled MFMA
    float* err6;      // per row: |v - dequantised row|_2 rounded up, plus the slack of the pass's fp32 / fixed-point arithmetic
    float* gerr6;     // its maximum per group
    float* gstep6;    // MX6_FIX_STEP for every group (the records are fixed-point: I8Bounds needs a step per group)
    float* err6h;     // per row: the same residual norm over the FIRST d / 2 columns only -- what the half-width pass multiplies
    float* gerr6h;    // its maximum per group
    // the int8 image once more, ROW-major (d bytes per row), for operands of at most ROWS8_MAX_ROWS rows -- scans: the chunk-major
    // rescan gathers 32 queries' k-steps per LDS-DMA instruction, and from the fragment tiles that is 64 different 128-byte lines per
    // instruction (a query's 16-byte units lie 512 bytes apart), from rows 32 lines that the block's other k-steps touch again
    unsigned char* rows8;
    size_t bytes;
};
constexpr int64_t ROWS8_MAX_ROWS = 131072;

// widths the int8 coarse pass exists for (match_coarse_pipe_kernel<d/32, false, true>; d = 128 has too few k-steps for the
// fragment ring)
inline bool i8_capable(int d) { return d == 256 || d == 384 || d == 512 || d == 640 || d == 768; }
// shapes the half-width pass (VFM_RECORDS_HALF) has a kernel for (every int8 width: the 64-queries-per-wave kernel for d <= 384
// with more than 2048 queries, the one-set kernel with four tiles per step elsewhere)
inline bool half_capable(int d, int64_t n) {
    (void)n;
    return i8_capable(d);
}
// widths the fp6 pass (VFM_RECORDS_MX6) has a kernel for: two resident query sets of d / 64 x 8 registers
inline bool mx6_width(int d) { return d == 256 || d == 384; }
// widths whose HALF-width pass has an fp6 kernel (d / 128 k-steps of queries in registers): the fp6 image exists for these
inline bool mx6_half_width(int d) { return d == 256 || d == 384 || d == 512 || d == 768; }
constexpr float MX6_FIX_STEP = 0.0009765625f;   // 2^-10: an fp6 record is ceil(score * 2^20) -- "integer score" x step x step
// ---------------------------------------------------------------------------------------------
// fp6 image (VFM_PREPARE_MX6), dense since round 4.  Lane l of v_mfma_scale_f32_32x32x64_f8f6f4 holds, for row l & 31 of a 32-row
// tile, the 32 columns 64 s + 32 (l >> 5) ... of k-step s as 32 consecutive 6-bit codes = 24 bytes, and one E8M0 scale per
// k-step.  A stored tile (KS = d / 64 k-steps):
//     [scale plane 0: 64 lanes x 8 B -- byte s of lane l = scale of k-step s < 8]
//     [k-step 0: plane A = 64 lanes x 16 B (code bytes 0 .. 15) | plane B = 64 lanes x 8 B (code bytes 16 .. 23)] [k-step 1] ...
//     [scale plane 1 (KS > 8 only): k-steps 8 ..]
// so a wave's fragment of a k-step is one ds_read_b128 at 16-byte stride + one ds_read_b64 at 8-byte stride -- both free of bank
// conflicts (MI355X_MICROARCH.md, LDS; round 3's 16-byte units with 8 spare bytes cost 35 % of the LDS cycles in conflicts and a
// third more bytes to stage) -- and the scales + the first KS / 2 k-steps, what the half-width pass reads, are a PREFIX of the tile.
// ---------------------------------------------------------------------------------------------
constexpr int MX6_SCALE_PLANE = 512;
constexpr int MX6_KSTEP_BYTES = 1536;
__host__ __device__ constexpr int mx6_tile_bytes(int ks) { return MX6_SCALE_PLANE + ks * MX6_KSTEP_BYTES + (ks > 8 ? MX6_SCALE_PLANE : 0); }
__host__ __device__ inline size_t mx6_code_a(int s, int lane) { return (size_t)MX6_SCALE_PLANE + (size_t)s * MX6_KSTEP_BYTES + (size_t)lane * 16; }
__host__ __device__ inline size_t mx6_code_b(int s, int lane) { return (size_t)MX6_SCALE_PLANE + (size_t)s * MX6_KSTEP_BYTES + 1024 + (size_t)lane * 8; }
__host__ __device__ inline size_t mx6_scale_at(int ks, int s, int lane) {
    return (s < 8 ? 0 : (size_t)MX6_SCALE_PLANE + (size_t)ks * MX6_KSTEP_BYTES) + (size_t)lane * 8 + (size_t)(s & 7);
}
// the fused form exists where the 64-queries-per-wave kernel runs and a map chunk collects several queries (chunk-major rescan)
inline int effective_records(int records, int d, int64_t n, int64_t m) {
    if (records == VFM_RECORDS_HALF_FUSED && !((d == 256 || d == 384) && n > 2048 && n >= 4 * ((m + CHUNK_ROWS - 1) / CHUNK_ROWS)))
        records = VFM_RECORDS_HALF;
    // (the pilot rescan is chunk-major: several queries per map chunk)
    if (records == VFM_RECORDS_MX6_PILOT && !(n >= 4 * ((m + CHUNK_ROWS - 1) / CHUNK_ROWS))) records = VFM_RECORDS_MX6;
    // (the fused full-width form: the chunk-major rescan behind it, like the other fused kinds)
    if (records == VFM_RECORDS_MX6_FUSED && !(n >= 4 * ((m + CHUNK_ROWS - 1) / CHUNK_ROWS))) records = VFM_RECORDS_MX6;
    if ((records == VFM_RECORDS_MX6 || records == VFM_RECORDS_MX6_PILOT || records == VFM_RECORDS_MX6_FUSED) && !(mx6_width(d) && n > 2048)) records = VFM_RECORDS_BEST;   // (the one-set kernels have no fp6 form)
    if (records == VFM_RECORDS_MX6_TOP2 && !(mx6_width(d) && n > 2048)) records = VFM_RECORDS_TOP2;
    if (records == VFM_RECORDS_MX6_HALF && !(mx6_half_width(d) && n > 2048)) records = VFM_RECORDS_BEST;   // (such operands carry no int8 half image)
    // the fused form needs the chunk-major rescan behind it (several queries per map chunk), like VFM_RECORDS_HALF_FUSED
    if (records == VFM_RECORDS_MX6_HALF_FUSED)
        records = !(mx6_half_width(d) && n > 2048) ? VFM_RECORDS_BEST
                  : (n >= 4 * ((m + CHUNK_ROWS - 1) / CHUNK_ROWS) ? VFM_RECORDS_MX6_HALF_FUSED : VFM_RECORDS_MX6_HALF);
    return (records == VFM_RECORDS_HALF && !half_capable(d, n)) ? VFM_RECORDS_BEST : records;
}

inline Prepared carve_prepared(void* p, int64_t rows, int d) {
    VfmCarver c(p);
    Prepared r;
    const int64_t rp = rows_padded(rows);
    r.inv = c.take<float>((size_t)rp);
    r.tiles = c.take<uint4>((size_t)rp / TILE_ROWS * (size_t)(d / 16) * 64);
    r.err = nullptr;
    r.gstep = r.gerr = nullptr;
    r.tiles8 = nullptr;
    r.tiles8h = nullptr;
    r.rest = r.grest = nullptr;
    r.tiles6 = nullptr;
    r.err6 = r.gerr6 = r.gstep6 = nullptr;
    r.err6h = r.gerr6h = nullptr;
    r.rows8 = nullptr;
    if (i8_capable(d)) {  // behind the fp16 image: the Euclidean path carves the same layout and ignores the rest
        r.err = c.take<float>((size_t)rp);
        r.gstep = c.take<float>((size_t)rp / I8_GROUP);
        r.gerr = c.take<float>((size_t)rp / I8_GROUP);
        r.tiles8 = c.take<uint4>((size_t)rp / TILE_ROWS * (size_t)(d / 32) * 64);
        // half-width pass (VFM_RECORDS_HALF): the int8 image of the first d / 2 columns as tiles of their own, the norm of
        // the OTHER half of every normalised row (rounded up) and its maximum per 128-row group
        r.tiles8h = c.take<uint4>((size_t)rp / TILE_ROWS * (size_t)(d / 64) * 64);
        r.rest = c.take<float>((size_t)rp);
        r.grest = c.take<float>((size_t)rp / I8_GROUP);
        if (mx6_half_width(d)) {   // the fp6 image: 24 bytes of codes per (row, 64 columns) + the block scales (mx6_tile_bytes)
            r.tiles6 = c.take<uint4>((size_t)rp / TILE_ROWS * (size_t)(mx6_tile_bytes(d / 64) / 16));
            r.err6 = c.take<float>((size_t)rp);
            r.gerr6 = c.take<float>((size_t)rp / I8_GROUP);
            r.gstep6 = c.take<float>((size_t)rp / I8_GROUP);
            r.err6h = c.take<float>((size_t)rp);
            r.gerr6h = c.take<float>((size_t)rp / I8_GROUP);
        }
        if (rp <= ROWS8_MAX_ROWS) r.rows8 = c.take<unsigned char>((size_t)rp * (size_t)d);
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
    unsigned* bin_cnt;  // int8 pass, best-score records: candidate queries per map chunk ...
    unsigned* hit_cnt;  // rows match_rescan_chunk_kernel appended to a query's list, one counter per 128-byte line (added to cand_cnt by
                        // match_rescan_close_kernel): the lists' own lengths sit 32 to a line, and a line is what the memory side serialises
    int* bins;          // ... and the queries themselves, bin_cap per chunk (match_rescan_chunk_kernel)
    int bin_cap;        // rescan_bin_cap(npad, chunks)
    unsigned long long* qbest;   // [npad] VFM_RECORDS_MX6_PILOT: the query's best chunk (CoarseArgs::qbest); inside the zeroed region
    float* cand_up;     // [npad][cap] upper bound of the exact score of the rows match_rescan_chunk_kernel appended (same positions as
                        // `cand`; other entries' slots are never read): match_refine_kernel drops rows below the list's best lower bound
    size_t bytes;
};

inline SearchWs carve_search(void* p, int64_t n, int64_t m) {
    VfmCarver c(p);
    SearchWs w;
    const int64_t npad = rows_padded(n), mpad = rows_padded(m);
    // the records of a coarse pass: per (chunk, query) 8 bytes -- or, for the fused fp6 half-width pass (which writes none), one slot
    // of MX6_SURV_SLOT_WORDS words per workgroup: (npad / 512 query blocks) x (at most min(64, chunks / 8) slices, at least one)
    {
        const size_t nch = (size_t)(mpad / CHUNK_ROWS);
        size_t slices = nch / 8 < 1 ? 1 : (nch / 8 > 64 ? 64 : nch / 8);
        if (slices < (nch + 254) / 255) slices = (nch + 254) / 255;   // (a slice's per-chunk constants fit the kernel's LDS table: <= 255 chunks)
        const size_t slots = ((size_t)npad / 512 + 1) * slices * (size_t)MX6_SURV_SLOT_WORDS;   // 4-byte words
        const size_t recs = nch * (size_t)npad;                                                  // 8-byte records
        w.partials = c.take<uint2>(recs > (slots + 1) / 2 ? recs : (slots + 1) / 2);
    }
    w.cand_cnt = c.take<int>((size_t)npad);
    w.cap = cand_cap(mpad);
    w.cand = c.take<unsigned>((size_t)npad * (size_t)w.cap);
    w.fb_list = c.take<int>((size_t)npad);
    // zeroed before every search by ONE memset (search_zero_bytes): [fb_count (64, padded to 256 B) | qmax (npad) |
    // rec_cnt (npad) | bin_cnt (chunks padded to 64, x BIN_CNT_STRIDE) | hit_cnt (npad x BIN_CNT_STRIDE)]; npad is a multiple of 256, so the arrays are contiguous under the
    // carver's 256-byte alignment
    w.fb_count = c.take<int>(64);
    w.qmax = c.take<unsigned>((size_t)npad);
    w.rec_cnt = c.take<unsigned>((size_t)npad);
    w.bin_cnt = c.take<unsigned>((size_t)((mpad / CHUNK_ROWS + 63) / 64 * 64) * BIN_CNT_STRIDE);
    w.hit_cnt = c.take<unsigned>((size_t)npad * BIN_CNT_STRIDE);
    w.qbest = c.take<unsigned long long>((size_t)npad);   // (the last array of the region search_zero_bytes covers)
    w.bin_cap = rescan_bin_cap(npad, mpad / CHUNK_ROWS);
    w.bins = c.take<int>((size_t)(mpad / CHUNK_ROWS) * (size_t)w.bin_cap);
    w.rcap = FILTER_LDS_ROWS;
    w.rec = c.take<uint2>((size_t)npad * (size_t)w.rcap);
    w.cand_up = c.take<float>((size_t)npad * (size_t)w.cap);
    w.bytes = c.used();
    return w;
}

inline size_t search_zero_bytes(int64_t n, int64_t m) {
    const int64_t npad = rows_padded(n), mpad = rows_padded(m);
    return 256 + 2 * (size_t)npad * sizeof(unsigned) + (size_t)((mpad / CHUNK_ROWS + 63) / 64 * 64) * BIN_CNT_STRIDE * sizeof(unsigned) +
           (size_t)npad * BIN_CNT_STRIDE * sizeof(unsigned) + (size_t)npad * sizeof(unsigned long long);
}

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

// Descriptor rows as the caller stores them (round 5, BASELINE.json configs[4] "fp16 descriptor storage"): fp32 -- the reference's layout
// behind VoxelHashMap.cpp:469-482 -- or fp16 (VFM_ROWS_F16: half the bytes of a resident map and of the preparation's read), every
// element widened to fp32 as it is loaded; from there on the arithmetic is the fp32 path's, operation for operation -- the oracle of an
// fp16 operand is the oracle of its widened rows.  A `const float*` converts implicitly (f16 = 0).
struct Rows {
    const void* p;
    int f16;
    __host__ __device__ Rows() : p(nullptr), f16(0) {}
    __host__ __device__ Rows(const float* x) : p(x), f16(0) {}
    __host__ __device__ Rows(const void* x, int is_f16) : p(x), f16(is_f16) {}
#if defined(__HIPCC__)
    typedef _Float16 rows_half4 __attribute__((ext_vector_type(4)));
    // four consecutive elements from element index e (a multiple of 4)
    __device__ __forceinline__ float4 ld4(int64_t e) const {
        if (f16) {
            const rows_half4 h = *reinterpret_cast<const rows_half4*>(static_cast<const _Float16*>(p) + e);
            return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        }
        return *reinterpret_cast<const float4*>(static_cast<const float*>(p) + e);
    }
    // the same, streamed (read-once rows of the preparation: do not displace the coarse pass' map slice from the L2)
    __device__ __forceinline__ float4 ld4_nt(int64_t e) const {
        if (f16) {
            const unsigned* pc = reinterpret_cast<const unsigned*>(static_cast<const _Float16*>(p) + e);
            unsigned w[2] = {__builtin_nontemporal_load(pc), __builtin_nontemporal_load(pc + 1)};
            rows_half4 h;
            __builtin_memcpy(&h, w, 8);
            return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        }
        const float* pc = static_cast<const float*>(p) + e;
        return make_float4(__builtin_nontemporal_load(pc), __builtin_nontemporal_load(pc + 1), __builtin_nontemporal_load(pc + 2),
                           __builtin_nontemporal_load(pc + 3));
    }
    __device__ __forceinline__ float ld1(int64_t e) const {
        return f16 ? (float)static_cast<const _Float16*>(p)[e] : static_cast<const float*>(p)[e];
    }
#endif
};

// experiment knobs and profiling hook (match_api.hip)
extern thread_local hipEvent_t g_prof_start, g_prof_stop;  // vfm_prof_arm: events around the next coarse launch of this thread

// which coarse pass / record kind a search takes (match_api.hip)
bool use_sparse(int d, int64_t n, int64_t m);
bool use_i8(int d, int64_t n, int64_t m, bool gated);
int coarse_qblock(int d);
int choose_slices(int nqb, int nchunks);
inline I8Bounds i8_bounds(const Prepared& Q, const Prepared& B, bool on, int top2 = 0) {
    return on ? I8Bounds{Q.err, Q.gstep, B.gstep, B.gerr, top2 == VFM_RECORDS_TOP2 ? 1 : 0} : I8Bounds{nullptr, nullptr, nullptr, nullptr, 0};
}
// bounds of the fp6 records (selection only: the rescans score the int8 image and use i8_bounds)
inline I8Bounds mx6_bounds(const Prepared& Q, const Prepared& B, int top2 = 0) { return I8Bounds{Q.err6, Q.gstep6, B.gstep6, B.gerr6, top2}; }
// ... of the half-width fp6 kinds: the residual norms over the columns the pass multiplies (tighter, and all a VFM_PREPARE_MX6_HALF
// operand carries: its full-width E is infinite)
inline I8Bounds mx6_bounds_half(const Prepared& Q, const Prepared& B) { return I8Bounds{Q.err6h, Q.gstep6, B.gstep6, B.gerr6h, 0}; }
CoarseArgs coarse_args(const Prepared& Q, const Prepared& B, const SearchWs& w, int64_t n, int64_t m, int qblock = QBLOCK);

// match_prep.hip
int do_prepare2(Rows x1, int64_t rows1, void* prepared1, Rows x2, int64_t rows2, void* prepared2, int d,
                hipStream_t st, bool want_f16 = true, int grid_mode = 0 /* VFM_PREPARE_DEFAULT */);
int do_prepare(const float* x, int64_t rows, int d, void* prepared, hipStream_t st);
// int8 image alone of ONE operand whose row r is x[perm[r]] (perm == NULL: identity) -- the Euclidean search prepares its map
// sorted by norm and the queries of its reverse direction gathered by the forward result
int do_prepare_perm(const float* x, int64_t rows, const int* perm, int d, void* prepared, hipStream_t st);
// match_coarse_f16.hip / match_coarse_i8.hip: launch the coarse kernel for arguments prepared by do_search_coarse
int launch_coarse_f16(const CoarseArgs& a, int d, hipStream_t st);
int launch_coarse_int8(CoarseArgs& a, int d, int64_t n, int records, hipStream_t st);
int launch_coarse_mx6(CoarseArgs& a, int d, bool top2, bool half, bool fuse, hipStream_t st);   // match_coarse_mx6.hip (fuse && !half: VFM_RECORDS_MX6_FUSED)
int mx6_survivor_slot_words();   // words per workgroup slot of the fused half-width pass (header + entries)
// match_api.hip
int do_search_coarse(const void* qprep, int64_t n, const void* bprep, int64_t m, int d, void* ws, hipStream_t st,
                     bool bias_from_map_inv = false, bool inner_product = false, bool gated = false, int records = 0,
                     float gate = -__builtin_inff());
// match_finish.hip
int do_search_finish(Rows q, const void* qprep, int64_t n, Rows b, const void* bprep, int64_t m, int d,
                     int64_t* idx_out, float* sim_out, void* ws, hipStream_t st, bool gated = false,
                     float gate = -__builtin_inff(), int records = 0);
int launch_select_dense(const SearchWs& w, const CoarseArgs& a, const float* qinv, int64_t n, hipStream_t st);
// candidate chunks (bins / lists of a selection kernel) -> candidate rows: query-major + chunk-major int8 rescans and the
// closing pass (over-long lists to the all-pairs fallback); l2.qn != NULL: the Euclidean hit test, qmax = float_key(qlow)
int launch_i8_rescans(const SearchWs& w, const CoarseArgs& a, const Prepared& Q, const Prepared& B, int64_t n, int64_t m, int d,
                      bool use_bins, L2Terms l2, hipStream_t st);
int probe_half_select(const void* qprep, int64_t n, const void* bprep, int64_t m, int d, void* ws, float gate, hipStream_t st);
int exact_ip_top1(const float* q, int64_t n, const float* b, int64_t m, int d, int64_t* idx_out, float* sim_out, void* ws,
                  hipStream_t st);

}  // namespace vfmm
