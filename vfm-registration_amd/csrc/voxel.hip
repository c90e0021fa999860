// voxel.hip -- voxel down-sampling / voxel-hash-map insertion on MI355X (gfx950), row F1.
//
// Replaces kiss_icp::VoxelDownsample (Preprocessing.cpp:50-137: first point of every voxel) and
// VoxelHashMap::AddPoints + Pointcloud*/GetVFMCorrespondences' dump of the map (VoxelHashMap.cpp:733-770,
// 640-676, 465; VoxelHashMap.hpp:55-62: at most max_points_per_voxel points per voxel, in insertion order).
//
// (1) WHICH points survive:   keep point i  <=>  fewer than K earlier points (j < i) fall into the same
//     voxel, voxel = trunc(xyz / voxel_size) per axis (Eigen cast<int>, Preprocessing.cpp:58).  Computed in
//     parallel and exactly: an open-addressing table of OWNER POINT INDICES (atomicCAS on 32 bits; a probe
//     compares the full int32 x 3 voxel of the owner, recomputed from its coordinates -- no packed key, so no
//     range limit and no aliasing), per-slot count and minimum index, then K-1 rounds of "smallest
//     not-yet-taken index" for the voxels that hold more than K points.
// (2) In WHICH ORDER they are emitted: the reference iterates a tsl::robin_map (Preprocessing.cpp:64-69,
//     VoxelHashMap.cpp:662-676), and the next voxelisation level / the map's row numbering depend on that
//     order (registration_node.py:399-414).  vfm_voxel_robin reproduces it: a robin-hood table with linear
//     probing is, cyclically, its keys sorted by home bucket, so the occupied bucket ranges ("clusters") follow
//     from ONE stable radix sort by `hash & mask` + a running maximum; clusters never interact, so one thread
//     per cluster replays the container's insertions (arrival order, tsl's swap rule: a displaced entry
//     leapfrogs entries of equal distance) inside its own window.  A map that grows (default-constructed
//     VoxelHashMap::map_) is the same step once per table generation, fed with the previous generation's
//     iteration order -- exactly what rehash_impl does.  Entries that wrap past the last bucket are handled by
//     re-running the generation in coordinates rotated to start at a bucket that is provably empty.
// HBM-bound integer work: coalesced SoA passes, atomics only on the (L2-resident) tables; hipCUB supplies the
// device-wide radix sort / scan / select primitives.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace {

constexpr int EMPTY_OWNER = -1;
constexpr int RH_DIST_LIMIT = 8192;  // tsl::detail_robin_hash::bucket_entry::DIST_FROM_IDEAL_BUCKET_LIMIT (v1.2.1)
constexpr int RH_MAX_CLUSTER = 4096;  // longest run of occupied buckets the per-cluster replay accepts
constexpr int SMALL_CAP = 512;       // generations up to 1024 buckets are replayed by one thread in LDS

struct Vox {
    int x, y, z;
};
__device__ __forceinline__ Vox voxel_of(const double* __restrict__ p, double vs) {
    Vox v;
    v.x = (int)(p[0] / vs);
    v.y = (int)(p[1] / vs);
    v.z = (int)(p[2] / vs);
    return v;
}
__device__ __forceinline__ unsigned slot_hash(Vox v) {  // table slot selection only (not the reference's hash)
    unsigned long long x = ((unsigned long long)(unsigned)v.x << 32) ^ ((unsigned long long)(unsigned)v.y << 16) ^
                           (unsigned long long)(unsigned)v.z;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    x ^= (unsigned long long)(unsigned)v.y * 0x9e3779b97f4a7c15ull;
    return (unsigned)(x >> 20);
}
// VoxelHash (Preprocessing.cpp:41-46 / VoxelHashMap.hpp:72-77): uint32 arithmetic, 20-bit mask
__device__ __forceinline__ unsigned reference_hash(Vox v, unsigned mul_y) {
    return ((1u << 20) - 1u) & (((unsigned)v.x * 73856093u) ^ ((unsigned)v.y * mul_y) ^ ((unsigned)v.z * 83492791u));
}

__global__ __launch_bounds__(256) void voxel_init_kernel(int* __restrict__ owner, int* __restrict__ tcount,
                                                         int* __restrict__ tmin, int64_t hsize) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= hsize) return;
    owner[s] = EMPTY_OWNER;
    tcount[s] = 0;
    tmin[s] = 0x7fffffff;
}

__global__ __launch_bounds__(256) void voxel_insert_kernel(const double* __restrict__ pts, int64_t n, int64_t stride,
                                                           double vs, int* __restrict__ owner, int* __restrict__ tcount,
                                                           int* __restrict__ tmin, int64_t hmask, int* __restrict__ slot_of) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Vox v = voxel_of(pts + i * stride, vs);
    int64_t s = (int64_t)slot_hash(v) & hmask;
    while (true) {
        const int prev = atomicCAS(owner + s, EMPTY_OWNER, (int)i);
        if (prev == EMPTY_OWNER) break;
        const Vox o = voxel_of(pts + (int64_t)prev * stride, vs);  // pts is immutable: no torn key
        if (o.x == v.x && o.y == v.y && o.z == v.z) break;
        s = (s + 1) & hmask;
    }
    slot_of[i] = (int)s;
    atomicAdd(tcount + s, 1);
    atomicMin(tmin + s, (int)i);
}

// state: 1 keep, 0 drop, 2 undecided (crowded voxel, K > 1); first[i] = 1 iff i is the first point of its voxel
__global__ __launch_bounds__(256) void voxel_classify_kernel(int64_t n, int K, const int* __restrict__ tcount,
                                                             const int* __restrict__ tmin,
                                                             const int* __restrict__ slot_of, uint8_t* __restrict__ state,
                                                             uint8_t* __restrict__ first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    const bool is_first = tmin[s] == (int)i;
    uint8_t st;
    if (tcount[s] <= K) st = 1;
    else if (is_first) st = 1;       // the first point of a crowded voxel is always kept
    else st = (K == 1) ? 0 : 2;
    state[i] = st;
    first[i] = is_first ? 1 : 0;
}

// one round: among the undecided points of every crowded voxel the smallest index is taken
__global__ __launch_bounds__(256) void voxel_round_min_kernel(int64_t n, const int* __restrict__ slot_of,
                                                              const uint8_t* __restrict__ state, int* __restrict__ tmin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || state[i] != 2) return;
    atomicMin(tmin + slot_of[i], (int)i);
}
__global__ __launch_bounds__(256) void voxel_round_reset_kernel(int64_t n, const int* __restrict__ slot_of,
                                                                const uint8_t* __restrict__ state, int* __restrict__ tmin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || state[i] != 2) return;
    tmin[slot_of[i]] = 0x7fffffff;  // benign race: every writer stores the same value
}
__global__ __launch_bounds__(256) void voxel_round_take_kernel(int64_t n, const int* __restrict__ slot_of,
                                                               uint8_t* __restrict__ state, const int* __restrict__ tmin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || state[i] != 2) return;
    if (tmin[slot_of[i]] == (int)i) state[i] = 1;
}
__global__ __launch_bounds__(256) void voxel_keepflag_kernel(int64_t n, const uint8_t* __restrict__ state,
                                                             uint8_t* __restrict__ keepflag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keepflag[i] = state[i] == 1 ? 1 : 0;
}

// ---------------------------------------------------------------------------------- robin_map order
// voxel v (first-appearance rank): reference hash + the table slot -> v map for the kept points of its block
__global__ __launch_bounds__(256) void robin_voxel_info_kernel(const double* __restrict__ pts, int64_t stride, double vs,
                                                               unsigned mul_y, const int64_t* __restrict__ vfirst,
                                                               const int64_t* __restrict__ nv_p,
                                                               const int* __restrict__ slot_of,
                                                               unsigned* __restrict__ vhash, int* __restrict__ slot_vid) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= *nv_p) return;
    const int64_t p = vfirst[v];
    vhash[v] = reference_hash(voxel_of(pts + p * stride, vs), mul_y);
    slot_vid[slot_of[p]] = (int)v;
}

// The first generations of a growing map (<= SMALL_CAP entries, <= 1024 buckets): the container itself,
// replayed by one thread with the table in LDS (tsl::robin_hash::insert_impl / rehash_impl, v1.2.1).
// info[0] = entries consumed, info[1] = bucket count reached, info[2] = max distance seen.
__global__ __launch_bounds__(64) void robin_small_kernel(const unsigned* __restrict__ vhash, int64_t nv, int* __restrict__ order,
                                                         int64_t* __restrict__ info) {
    __shared__ short dist_a[2 * SMALL_CAP], dist_b[2 * SMALL_CAP];
    __shared__ int id_a[2 * SMALL_CAP], id_b[2 * SMALL_CAP];
    if (threadIdx.x != 0) return;
    short* dist = dist_a;
    int* id = id_a;
    short* dist2 = dist_b;
    int* id2 = id_b;
    int B = 0, nb = 0, maxd = 0;
    const int todo = (int)(nv < SMALL_CAP ? nv : SMALL_CAP);
    auto place = [&](short* D, int* I, int mask, int ib, int d, int v) {  // insert_value_on_rehash == insert + swap chain
        for (;;) {
            if (d > D[ib]) {
                if (D[ib] < 0) { D[ib] = (short)d; I[ib] = v; if (d > maxd) maxd = d; return; }
                const int td = D[ib], tv = I[ib];
                D[ib] = (short)d; I[ib] = v;
                if (d > maxd) maxd = d;
                d = td; v = tv;
            }
            d++;
            ib = (ib + 1) & mask;
        }
    };
    for (int v = 0; v < todo; ++v) {
        if (nb >= B / 2) {  // size() >= load_threshold -> rehash_impl(next_bucket_count())
            const int B2 = B ? 2 * B : 2;
            for (int b = 0; b < B2; ++b) dist2[b] = -1;
            for (int b = 0; b < B; ++b)
                if (dist[b] >= 0) place(dist2, id2, B2 - 1, (int)(vhash[id[b]] & (unsigned)(B2 - 1)), 0, id[b]);
            short* td = dist; dist = dist2; dist2 = td;
            int* ti = id; id = id2; id2 = ti;
            B = B2;
        }
        place(dist, id, B - 1, (int)(vhash[v] & (unsigned)(B - 1)), 0, v);
        nb++;
    }
    int k = 0;
    for (int b = 0; b < B; ++b)
        if (dist[b] >= 0) order[k++] = id[b];
    info[0] = todo;
    info[1] = B;
    info[2] = maxd;
}

// generation input: seq = [previous iteration order (n_prev entries, already in seq) ++ voxels s0 .. s0+n_new-1]
__global__ __launch_bounds__(256) void robin_append_kernel(int* __restrict__ seq, int64_t n_prev, int64_t s0, int64_t n_new) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_new) seq[n_prev + i] = (int)(s0 + i);
}
__global__ __launch_bounds__(256) void robin_keys_kernel(const int* __restrict__ seq, int64_t m, const unsigned* __restrict__ vhash,
                                                         unsigned mask, unsigned z, unsigned* __restrict__ key,
                                                         int* __restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    key[i] = (vhash[seq[i]] - z) & mask;
    pos[i] = (int)i;
}
// d_i = home_i - i over the home-sorted entries; bucket of entry i = i + running_max(d)
__global__ __launch_bounds__(256) void robin_delta_kernel(const unsigned* __restrict__ key_s, int64_t m, int* __restrict__ d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < m) d[i] = (int)key_s[i] - (int)i;
}
// a cluster starts where the running maximum strictly increases
__global__ __launch_bounds__(256) void robin_flag_kernel(const int* __restrict__ d, const int* __restrict__ cm, int64_t m,
                                                         int* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < m) flag[i] = (i == 0 || d[i] > cm[i - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void robin_cluster_kernel(const int* __restrict__ flag, const int* __restrict__ cidx,
                                                            const unsigned* __restrict__ key_s, const int* __restrict__ pos_s,
                                                            int64_t m, unsigned* __restrict__ cl_of_pos,
                                                            int* __restrict__ cl_start, int* __restrict__ cl_base,
                                                            int* __restrict__ tab_dist) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int c = cidx[i] - 1;
    cl_of_pos[pos_s[i]] = (unsigned)c;
    tab_dist[i] = -1;
    if (flag[i]) {
        cl_start[c] = (int)i;
        cl_base[c] = (int)key_s[i];
    }
    if (i == m - 1) cl_start[c + 1] = (int)m;
}
// longest cluster of the generation (geninfo[4]): a saturated table (20-bit VoxelHash with ~2^20 voxels) has clusters of
// thousands of entries, which one thread per cluster would replay for seconds -- the host refuses those instead
__global__ __launch_bounds__(256) void robin_maxlen_kernel(const int* __restrict__ cl_start, const int* __restrict__ cidx, int64_t m,
                                                           int64_t* __restrict__ geninfo) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cidx[m - 1]) return;
    const int len = cl_start[c + 1] - cl_start[c];
    if (len > 64) atomicMax((unsigned long long*)(geninfo + 4), (unsigned long long)len);
}
// wrap analysis (one thread): w = entries whose bucket would be >= B; z = a bucket that stays empty once the w
// wrapped entries have filled the first w free buckets; r = first entry whose rotated bucket is >= B - z_used.
// geninfo: [0] w, [1] z, [2] rotation r for the CURRENT coordinates (z_used), [3] max dist (atomicMax by replay)
__global__ void robin_wrap_kernel(const int* __restrict__ cm, int64_t m, int64_t B, int64_t z_used, int64_t* __restrict__ geninfo) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // entries i with i + cm[i] >= B form a suffix (i + cm[i] is increasing)
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (mid + (int64_t)cm[mid] >= B) hi = mid; else lo = mid + 1;
    }
    const int64_t w = m - lo;
    int64_t z = 0;
    if (w > 0) {  // smallest i with cm[i] >= w + 1: the bucket just below its cluster is the cm[i]-th free one
        int64_t a = 0, b = m;
        while (a < b) {
            const int64_t mid = (a + b) >> 1;
            if ((int64_t)cm[mid] >= w + 1) b = mid; else a = mid + 1;
        }
        z = (a < m) ? a + (int64_t)cm[a] - 1 : -1;
    }
    int64_t r = 0;
    if (z_used > 0) {  // actual bucket = (rotated bucket + z_used) mod B: iteration starts at rotated bucket B - z_used
        int64_t a = 0, b = m;
        while (a < b) {
            const int64_t mid = (a + b) >> 1;
            if (mid + (int64_t)cm[mid] >= B - z_used) b = mid; else a = mid + 1;
        }
        r = a;
    }
    geninfo[0] = w;
    geninfo[1] = z;
    geninfo[2] = r;
    geninfo[3] = 0;
    geninfo[4] = 0;
}
// one thread per cluster replays the insertions of its entries (arrival order) inside its own window
__global__ __launch_bounds__(64) void robin_replay_kernel(const int* __restrict__ cl_start, const int* __restrict__ cl_base,
                                                          const int* __restrict__ cidx, int64_t m,
                                                          const int* __restrict__ arrivals, const unsigned* __restrict__ vhash,
                                                          unsigned mask, unsigned z, int* __restrict__ tab_dist,
                                                          int* __restrict__ tab_id, int64_t* __restrict__ geninfo) {
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t ncl = cidx[m - 1];
    if (c >= ncl) return;
    const int i0 = cl_start[c], L = cl_start[c + 1] - i0, base = cl_base[c];
    if (L > RH_MAX_CLUSTER) return;   // (one thread replays a cluster: the host rejects such a table from geninfo[4], in the same read-back)
    int* D = tab_dist + i0;
    int* I = tab_id + i0;
    int maxd = 0;
    for (int t = 0; t < L; ++t) {
        int v = arrivals[i0 + t];
        int ib = (int)((vhash[v] - z) & mask) - base;
        int d = 0;
        for (;;) {  // tsl insert_value_on_rehash (== insert_impl + insert_value_impl for an absent key)
            const int rd = D[ib];
            if (d > rd) {
                if (rd < 0) { D[ib] = d; I[ib] = v; if (d > maxd) maxd = d; break; }
                const int tv = I[ib];
                D[ib] = d; I[ib] = v;
                if (d > maxd) maxd = d;
                d = rd; v = tv;
            }
            d++;
            ib++;  // never leaves [0, L): the window is the cluster's final extent
        }
    }
    if (maxd > 0) atomicMax((unsigned long long*)(geninfo + 3), (unsigned long long)maxd);
}
__global__ __launch_bounds__(256) void robin_emit_kernel(const int* __restrict__ tab_id, int64_t m, const int64_t* __restrict__ geninfo,
                                                         int* __restrict__ order) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    int64_t j = i + geninfo[2];
    if (j >= m) j -= m;
    order[i] = tab_id[j];
}
// final assembly
__global__ __launch_bounds__(256) void robin_rank_kernel(const int* __restrict__ order, int64_t nv, unsigned* __restrict__ vrank) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nv) vrank[order[i]] = (unsigned)i;
}
__global__ __launch_bounds__(256) void robin_out1_kernel(const int* __restrict__ order, int64_t nv, const int64_t* __restrict__ vfirst,
                                                         int64_t* __restrict__ keep_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nv) keep_out[i] = vfirst[order[i]];
}
__global__ __launch_bounds__(256) void robin_pointkey_kernel(const int64_t* __restrict__ kept, int64_t nk, const int* __restrict__ slot_of,
                                                             const int* __restrict__ slot_vid, const unsigned* __restrict__ vrank,
                                                             unsigned* __restrict__ pkey) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < nk) pkey[k] = vrank[slot_vid[slot_of[kept[k]]]];
}

struct VoxelWs {
    int* owner;
    int* tcount;
    int* tmin;
    int* slot_of;
    uint8_t* state;
    uint8_t* first;
    uint8_t* keepflag;
    int64_t hsize;
    // robin part
    int64_t* kept;     // [n] kept point indices, ascending
    int64_t* vfirst;   // [n] first point of voxel v
    int64_t* counts;   // [0] nk, [1] nv
    int64_t* geninfo;  // [4]
    int64_t* smallinfo;  // [4]
    unsigned* vhash;   // [n]
    int* slot_vid;     // [hsize]
    int* seq_a;        // [n] iteration order / generation input (ping)
    int* seq_b;        // [n] (pong)
    unsigned* key;     // [n]
    unsigned* key_s;   // [n]
    int* pos;          // [n]
    int* pos_s;        // [n]
    int* d;            // [n]
    int* cm;           // [n]
    int* flag;         // [n]
    int* cidx;         // [n]
    unsigned* cl_of_pos;  // [n]
    unsigned* cl_sorted;  // [n]
    int* arrivals;     // [n]
    int* cl_start;     // [n+1]
    int* cl_base;      // [n]
    int* tab_dist;     // [n]
    int* tab_id;       // [n]
    void* cub;         // hipCUB temporary storage
    size_t cub_bytes;
    size_t bytes;
};

size_t cub_temp_bytes(int64_t n) {
    const int ni = (int)(n > 0 ? n : 1);
    size_t best = 0, b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (unsigned*)nullptr, (unsigned*)nullptr, (int*)nullptr, (int*)nullptr, ni, 0, 32);
    best = b > best ? b : best;
    b = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (unsigned*)nullptr, (unsigned*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr, ni, 0, 32);
    best = b > best ? b : best;
    b = 0;
    (void)hipcub::DeviceScan::InclusiveScan(nullptr, b, (int*)nullptr, (int*)nullptr, hipcub::Max(), ni);
    best = b > best ? b : best;
    b = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, b, (int*)nullptr, (int*)nullptr, ni);
    best = b > best ? b : best;
    b = 0;
    hipcub::CountingInputIterator<int64_t> it(0);
    (void)hipcub::DeviceSelect::Flagged(nullptr, b, it, (uint8_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr, ni);
    best = b > best ? b : best;
    return best + 1024;
}

inline VoxelWs carve_voxel(void* p, int64_t n, bool robin) {
    VfmCarver c(p);
    VoxelWs w{};
    const size_t nn = (size_t)(n > 0 ? n : 1);
    int64_t h = 1024;
    while (h < 2 * n) h <<= 1;
    w.hsize = h;
    w.owner = c.take<int>((size_t)h);
    w.tcount = c.take<int>((size_t)h);
    w.tmin = c.take<int>((size_t)h);
    w.slot_of = c.take<int>(nn);
    w.state = c.take<uint8_t>(nn);
    w.first = c.take<uint8_t>(nn);
    w.keepflag = c.take<uint8_t>(nn);
    w.kept = c.take<int64_t>(nn);
    w.counts = c.take<int64_t>(4);
    if (robin) {
        w.vfirst = c.take<int64_t>(nn);
        w.geninfo = c.take<int64_t>(8);
        w.smallinfo = c.take<int64_t>(4);
        w.vhash = c.take<unsigned>(nn);
        w.slot_vid = c.take<int>((size_t)h);
        w.seq_a = c.take<int>(nn);
        w.seq_b = c.take<int>(nn);
        w.key = c.take<unsigned>(nn);
        w.key_s = c.take<unsigned>(nn);
        w.pos = c.take<int>(nn);
        w.pos_s = c.take<int>(nn);
        w.d = c.take<int>(nn);
        w.cm = c.take<int>(nn);
        w.flag = c.take<int>(nn);
        w.cidx = c.take<int>(nn);
        w.cl_of_pos = c.take<unsigned>(nn);
        w.cl_sorted = c.take<unsigned>(nn);
        w.arrivals = c.take<int>(nn);
        w.cl_start = c.take<int>(nn + 1);
        w.cl_base = c.take<int>(nn);
        w.tab_dist = c.take<int>(nn);
        w.tab_id = c.take<int>(nn);
    }
    w.cub_bytes = (p != nullptr || true) ? cub_temp_bytes(n) : 0;
    w.cub = c.take<unsigned char>(w.cub_bytes);
    w.bytes = c.used();
    return w;
}

inline unsigned blocks_of(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per > 0 ? (n + per - 1) / per : 1); }

// step (1): state[i] == 1 for the survivors, first[i] for the first point of every voxel
int select_first_k(const double* pts, int64_t n, int64_t stride, double voxel_size, int32_t K, const VoxelWs& w, hipStream_t st,
                   bool want_keepflag = true) {
    const unsigned gb = blocks_of(n);
    hipLaunchKernelGGL(voxel_init_kernel, dim3(blocks_of(w.hsize)), dim3(256), 0, st, w.owner, w.tcount, w.tmin, w.hsize);
    if (n > 0) {
        hipLaunchKernelGGL(voxel_insert_kernel, dim3(gb), dim3(256), 0, st, pts, n, stride, voxel_size, w.owner, w.tcount,
                           w.tmin, w.hsize - 1, w.slot_of);
        hipLaunchKernelGGL(voxel_classify_kernel, dim3(gb), dim3(256), 0, st, n, (int)K, w.tcount, w.tmin, w.slot_of,
                           w.state, w.first);
        for (int r = 1; r < K; ++r) {  // rounds 2..K: next smallest undecided index per crowded voxel
            hipLaunchKernelGGL(voxel_round_reset_kernel, dim3(gb), dim3(256), 0, st, n, w.slot_of, w.state, w.tmin);
            hipLaunchKernelGGL(voxel_round_min_kernel, dim3(gb), dim3(256), 0, st, n, w.slot_of, w.state, w.tmin);
            hipLaunchKernelGGL(voxel_round_take_kernel, dim3(gb), dim3(256), 0, st, n, w.slot_of, w.state, w.tmin);
        }
        if (want_keepflag) hipLaunchKernelGGL(voxel_keepflag_kernel, dim3(gb), dim3(256), 0, st, n, w.state, w.keepflag);
    }
    return VFM_OK;
}

}  // namespace

VFM_EXPORT size_t vfm_voxel_first_workspace_bytes(int64_t n) { return carve_voxel(nullptr, n, false).bytes; }

VFM_EXPORT int vfm_voxel_first(const double* pts, int64_t n, int64_t stride, double voxel_size, int32_t max_per_voxel,
                               int64_t* keep_out, int64_t* count_out, void* ws, size_t ws_bytes, vfm_stream_t stream) {
    VFM_CHECK_ARG(pts && keep_out && count_out && ws && n >= 0 && stride >= 3, "voxel_first: bad arguments");
    VFM_CHECK_ARG(voxel_size > 0.0 && max_per_voxel >= 1 && n < (1ll << 31), "voxel_first: bad voxel_size / cap / n");
    if (ws_bytes < vfm_voxel_first_workspace_bytes(n)) return vfm_fail(VFM_EWORKSPACE, "voxel_first: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    VoxelWs w = carve_voxel(ws, n, false);
    if (n == 0) {
        VFM_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int64_t), st));
        return VFM_OK;
    }
    select_first_k(pts, n, stride, voxel_size, max_per_voxel, w, st);
    hipcub::CountingInputIterator<int64_t> it(0);
    size_t tb = w.cub_bytes;
    VFM_CHECK_HIP(hipcub::DeviceSelect::Flagged(w.cub, tb, it, w.keepflag, keep_out, count_out, (int)n, st));
    VFM_CHECK_LAUNCH("voxel_first kernels");
    return VFM_OK;
}

VFM_EXPORT size_t vfm_voxel_robin_workspace_bytes(int64_t n) { return carve_voxel(nullptr, n, true).bytes; }

VFM_EXPORT int vfm_voxel_robin(const double* pts, int64_t n, int64_t stride, double voxel_size, int32_t max_per_voxel,
                               uint32_t hash_mul_y, int64_t reserve_n, int64_t* keep_out, int64_t* count_out,
                               int64_t* info_host, void* ws, size_t ws_bytes, vfm_stream_t stream) {
    VFM_CHECK_ARG(pts && keep_out && count_out && ws && n >= 0 && stride >= 3, "voxel_robin: bad arguments");
    VFM_CHECK_ARG(voxel_size > 0.0 && max_per_voxel >= 1 && n < (1ll << 30), "voxel_robin: bad voxel_size / cap / n");
    if (ws_bytes < vfm_voxel_robin_workspace_bytes(n)) return vfm_fail(VFM_EWORKSPACE, "voxel_robin: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    VoxelWs w = carve_voxel(ws, n, true);
    if (info_host) info_host[0] = info_host[1] = info_host[2] = info_host[3] = 0;
    if (n == 0) {
        VFM_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int64_t), st));
        return VFM_OK;
    }
    const int K = max_per_voxel;
    select_first_k(pts, n, stride, voxel_size, K, w, st, K > 1);
    hipcub::CountingInputIterator<int64_t> it(0);
    size_t tb = w.cub_bytes;
    // (one point per voxel: the kept points ARE the voxels' first points -- one compaction instead of two, and nk = nv)
    if (K > 1) VFM_CHECK_HIP(hipcub::DeviceSelect::Flagged(w.cub, tb, it, w.keepflag, w.kept, w.counts + 0, (int)n, st));
    tb = w.cub_bytes;
    VFM_CHECK_HIP(hipcub::DeviceSelect::Flagged(w.cub, tb, it, w.first, w.vfirst, w.counts + 1, (int)n, st));

    hipLaunchKernelGGL(robin_voxel_info_kernel, dim3(blocks_of(n)), dim3(256), 0, st, pts, stride, voxel_size, hash_mul_y,
                       w.vfirst, w.counts + 1, w.slot_of, w.vhash, w.slot_vid);
    // the generation sizes depend on the number of voxels: one 16-byte read-back (this entry point synchronises)
    int64_t counts_h[2];
    VFM_CHECK_HIP(hipMemcpyAsync(counts_h, w.counts, sizeof(counts_h), hipMemcpyDeviceToHost, st));
    VFM_CHECK_HIP(hipStreamSynchronize(st));
    const int64_t nv = counts_h[1], nk = K == 1 ? nv : counts_h[0];

    // tsl::robin_map state: bucket count B, s entries inserted, current iteration order in `cur`
    int64_t B = 0, s = 0, max_dist = 0, n_wrapped_gens = 0;
    int* cur = w.seq_a;
    int* nxt = w.seq_b;
    if (reserve_n >= 0) {  // reserve(n): rehash(size_t(ceil(float(n) / 0.5f))), rounded up to a power of two
        const float c = ceilf((float)reserve_n / 0.5f);
        int64_t want = (int64_t)c;
        B = 0;
        if (want > 0) {
            B = 1;
            while (B < want) B <<= 1;
        }
    } else if (nv > 0) {
        hipLaunchKernelGGL(robin_small_kernel, dim3(1), dim3(64), 0, st, w.vhash, nv, cur, w.smallinfo);
        int64_t si[3];
        VFM_CHECK_HIP(hipMemcpyAsync(si, w.smallinfo, sizeof(si), hipMemcpyDeviceToHost, st));
        VFM_CHECK_HIP(hipStreamSynchronize(st));
        s = si[0];
        B = si[1];
        max_dist = si[2];
    }
    while (s < nv) {
        // load_threshold = size_t(float(B) * 0.5f); an insert with size() >= threshold first doubles the table
        int64_t thr = (int64_t)((float)B * 0.5f);
        if (s >= thr) {
            B = B ? 2 * B : 2;
            thr = (int64_t)((float)B * 0.5f);
        }
        const int64_t n_new = (nv < thr ? nv : thr) - s;  // entries that fit before the next doubling
        const int64_t m = s + n_new;
        VFM_CHECK_ARG(B <= (1ll << 31), "voxel_robin: bucket count beyond 2^31");
        const unsigned mask = (unsigned)(B - 1);
        int bits = 1;
        while ((1ll << bits) < B) ++bits;
        if (bits > 20) bits = 20;  // VoxelHash is masked to 20 bits
        hipLaunchKernelGGL(robin_append_kernel, dim3(blocks_of(n_new)), dim3(256), 0, st, cur, s, s, n_new);
        unsigned z = 0;
        for (int pass = 0; pass < 2; ++pass) {
            const unsigned gm = blocks_of(m);
            hipLaunchKernelGGL(robin_keys_kernel, dim3(gm), dim3(256), 0, st, cur, m, w.vhash, mask, z, w.key, w.pos);
            // with z != 0 the rotated homes span all of [0, B): sort on every bit of the mask
            int sort_bits = 1;
            while ((1ll << sort_bits) < B) ++sort_bits;
            if (z == 0 && sort_bits > bits) sort_bits = bits;
            tb = w.cub_bytes;
            VFM_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(w.cub, tb, w.key, w.key_s, w.pos, w.pos_s, (int)m, 0, sort_bits, st));
            hipLaunchKernelGGL(robin_delta_kernel, dim3(gm), dim3(256), 0, st, w.key_s, m, w.d);
            tb = w.cub_bytes;
            VFM_CHECK_HIP(hipcub::DeviceScan::InclusiveScan(w.cub, tb, w.d, w.cm, hipcub::Max(), (int)m, st));
            hipLaunchKernelGGL(robin_flag_kernel, dim3(gm), dim3(256), 0, st, w.d, w.cm, m, w.flag);
            tb = w.cub_bytes;
            VFM_CHECK_HIP(hipcub::DeviceScan::InclusiveSum(w.cub, tb, w.flag, w.cidx, (int)m, st));
            hipLaunchKernelGGL(robin_cluster_kernel, dim3(gm), dim3(256), 0, st, w.flag, w.cidx, w.key_s, w.pos_s, m,
                               w.cl_of_pos, w.cl_start, w.cl_base, w.tab_dist);
            hipLaunchKernelGGL(robin_wrap_kernel, dim3(1), dim3(1), 0, st, w.cm, m, B, (int64_t)z, w.geninfo);
            hipLaunchKernelGGL(robin_maxlen_kernel, dim3(gm), dim3(256), 0, st, w.cl_start, w.cidx, m, w.geninfo);
            int cbits = 1;
            while ((1ll << cbits) < m) ++cbits;
            tb = w.cub_bytes;
            VFM_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(w.cub, tb, w.cl_of_pos, w.cl_sorted, cur, w.arrivals, (int)m, 0,
                                                            cbits, st));
            hipLaunchKernelGGL(robin_replay_kernel, dim3(blocks_of(m, 64)), dim3(64), 0, st, w.cl_start, w.cl_base, w.cidx, m,
                               w.arrivals, w.vhash, mask, z, w.tab_dist, w.tab_id, w.geninfo);
            int64_t gi[5];   // one read-back per pass: wrap analysis, probe distance and the longest run (the replay skips runs beyond the limit)
            VFM_CHECK_HIP(hipMemcpyAsync(gi, w.geninfo, sizeof(gi), hipMemcpyDeviceToHost, st));
            VFM_CHECK_HIP(hipStreamSynchronize(st));
            if (gi[4] > RH_MAX_CLUSTER)
                return vfm_fail(VFM_EINVAL, "voxel_robin: a run of %lld occupied buckets -- the reference container's 20-bit "
                                "VoxelHash is saturated (%lld voxels in %lld buckets); not reproduced", (long long)gi[4],
                                (long long)m, (long long)B);
            if (gi[0] > 0) {  // entries wrapped past the last bucket: redo in rotated coordinates
                if (pass == 1 || gi[1] < 0) return vfm_fail(VFM_EINVAL, "voxel_robin: wrap analysis failed (table saturated)");
                z = (unsigned)gi[1];
                ++n_wrapped_gens;
                continue;
            }
            if (gi[3] > max_dist) max_dist = gi[3];
            break;
        }
        hipLaunchKernelGGL(robin_emit_kernel, dim3(blocks_of(m)), dim3(256), 0, st, w.tab_id, m, w.geninfo, nxt);
        int* t = cur; cur = nxt; nxt = t;
        s = m;
    }
    if (max_dist > RH_DIST_LIMIT)
        return vfm_fail(VFM_EINVAL, "voxel_robin: probe distance %lld exceeds tsl::robin_map's limit %d -- the reference "
                        "container would keep doubling here (20-bit VoxelHash saturated); not reproduced",
                        (long long)max_dist, RH_DIST_LIMIT);
    // final assembly: voxels in iteration order, the points of a voxel in insertion order
    if (nv > 0) {
        if (K == 1) {
            hipLaunchKernelGGL(robin_out1_kernel, dim3(blocks_of(nv)), dim3(256), 0, st, cur, nv, w.vfirst, keep_out);
        } else {
            unsigned* vrank = w.key;
            unsigned* pkey = w.key_s;
            hipLaunchKernelGGL(robin_rank_kernel, dim3(blocks_of(nv)), dim3(256), 0, st, cur, nv, vrank);
            hipLaunchKernelGGL(robin_pointkey_kernel, dim3(blocks_of(nk)), dim3(256), 0, st, w.kept, nk, w.slot_of, w.slot_vid,
                               vrank, pkey);
            int vbits = 1;
            while ((1ll << vbits) < nv) ++vbits;
            tb = w.cub_bytes;
            VFM_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(w.cub, tb, pkey, w.cl_of_pos, w.kept, keep_out, (int)nk, 0, vbits, st));
        }
    }
    VFM_CHECK_HIP(hipMemcpyAsync(count_out, w.counts + (K == 1 ? 1 : 0), sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    VFM_CHECK_LAUNCH("voxel_robin kernels");
    if (info_host) {
        info_host[0] = B;
        info_host[1] = nv;
        info_host[2] = max_dist;
        info_host[3] = n_wrapped_gens;
    }
    return VFM_OK;
}
