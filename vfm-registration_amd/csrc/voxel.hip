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
#include <atomic>
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
// Round 5: the same replay without the second radix sort in front of it and without a global-memory round trip per probe.  A cluster's
// members are a contiguous range of the HOME-sorted entries (pos_s = their positions in the generation's input sequence `cur`); their
// arrival order is ascending position, so the thread sorts its own range (insertion sort: clusters of a half-empty table hold one to a
// few entries) instead of the whole generation going through a radix sort by (cluster, arrival) -- ~8 launches per generation.  Clusters
// of up to REPLAY_L entries are replayed in a per-thread slice of the LDS (positions, voxels, homes fetched with independent loads first;
// the old kernel's chain of dependent global loads per probe was 30 us per generation at 20 000 voxels); longer ones in place, in global
// memory, as before.  Same insertions in the same order: the same table.
constexpr int REPLAY_L = 24;
__global__ __launch_bounds__(64) void robin_replay2_kernel(const int* __restrict__ cl_start, const int* __restrict__ cl_base,
                                                           const int* __restrict__ cidx, int64_t m, int* __restrict__ pos_s,
                                                           const int* __restrict__ cur, const unsigned* __restrict__ vhash,
                                                           unsigned mask, unsigned z, int* __restrict__ tab_dist,
                                                           int* __restrict__ tab_id, int64_t* __restrict__ geninfo) {
    __shared__ int sP[REPLAY_L][64], sH[REPLAY_L][64], sD[REPLAY_L][64], sI[REPLAY_L][64];   // [slot][thread]: conflict-free
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t ncl = cidx[m - 1];
    if (c >= ncl) return;
    const int tx = threadIdx.x;
    const int i0 = cl_start[c], L = cl_start[c + 1] - i0, base = cl_base[c];
    if (L > RH_MAX_CLUSTER) return;   // (the host rejects such a table from geninfo[4], in the same read-back)
    int maxd = 0;
    if (L <= REPLAY_L) {
        for (int t = 0; t < L; ++t) sP[t][tx] = pos_s[i0 + t];
        for (int x = 1; x < L; ++x) {   // arrival order = ascending position in `cur`
            const int p = sP[x][tx];
            int y = x - 1;
            while (y >= 0 && sP[y][tx] > p) { sP[y + 1][tx] = sP[y][tx]; --y; }
            sP[y + 1][tx] = p;
        }
        for (int t = 0; t < L; ++t) sP[t][tx] = cur[sP[t][tx]];                                   // positions -> voxels
        for (int t = 0; t < L; ++t) sH[t][tx] = (int)((vhash[sP[t][tx]] - z) & mask) - base;     // their homes inside the window
        for (int t = 0; t < L; ++t) sD[t][tx] = -1;
        for (int t = 0; t < L; ++t) {
            int v = sP[t][tx], ib = sH[t][tx], d = 0;
            for (;;) {  // tsl insert_value_on_rehash (== insert_impl + insert_value_impl for an absent key)
                const int rd = sD[ib][tx];
                if (d > rd) {
                    if (rd < 0) { sD[ib][tx] = d; sI[ib][tx] = v; if (d > maxd) maxd = d; break; }
                    const int tv = sI[ib][tx];
                    sD[ib][tx] = d; sI[ib][tx] = v;
                    if (d > maxd) maxd = d;
                    d = rd; v = tv;
                }
                d++;
                ib++;  // never leaves [0, L): the window is the cluster's final extent
            }
        }
        for (int t = 0; t < L; ++t) {
            tab_dist[i0 + t] = sD[t][tx];
            tab_id[i0 + t] = sI[t][tx];
        }
    } else {
        int* P = pos_s + i0;
        for (int x = 1; x < L; ++x) {
            const int p = P[x];
            int y = x - 1;
            while (y >= 0 && P[y] > p) { P[y + 1] = P[y]; --y; }
            P[y + 1] = p;
        }
        int* D = tab_dist + i0;
        int* I = tab_id + i0;
        for (int t = 0; t < L; ++t) {
            int v = cur[P[t]];
            int ib = (int)((vhash[v] - z) & mask) - base;
            int d = 0;
            for (;;) {
                const int rd = D[ib];
                if (d > rd) {
                    if (rd < 0) { D[ib] = d; I[ib] = v; if (d > maxd) maxd = d; break; }
                    const int tv = I[ib];
                    D[ib] = d; I[ib] = v;
                    if (d > maxd) maxd = d;
                    d = rd; v = tv;
                }
                d++;
                ib++;
            }
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

// ---------------------------------------------------------------------------------- one launch for VoxelDownsample-sized clouds
// Round 5.  The reference-shaped call chains three VoxelDownsample()s of 10^3 .. 10^5 points
// (registration_node.py:399-414); through the general path above each is ~40 dependent launches and two read-backs (0.3 ms: launch
// latency, not work).  For ONE generation of a reserved table (reserve(n): B = 2^ceil(log2 2n) buckets, no rehash), one point per
// voxel, voxel_robin_grid_kernel does the whole of it in one launch: up to 256 workgroups of 256 threads, all resident at once, walk
// through the phases together, a grid-wide barrier (agent-scope release -> one atomic on an arrival counter -> spin -> acquire: ~3 us)
// where the general path has a launch boundary (~8 us) and a device-wide scan as "every workgroup's total, barrier, 256-entry prefix":
//   0  tables and histogram cleared;  A  the first-point table (atomicCAS / atomicMin, as voxel_insert_kernel);
//   B  first points in ascending order (grid scan) = the voxels in arrival order, their 20-bit VoxelHash, home-bucket histogram;
//   C  the voxels counting-sorted by home bucket (grid scan of the histogram, scatter) -- ties in any order: the replay below
//      orders a cluster's members by arrival itself;
//   D  bucket of sorted entry i = i + running max (home_i - i) (grid max-scan): clusters, their windows, the entries that would wrap
//      past the last bucket (then C and D once more in coordinates rotated to a bucket that stays empty -- robin_wrap_kernel's rule);
//   E  one thread per cluster: members sorted by arrival (insertion sort: clusters of a half-empty table are a few entries), then the
//      container's insertions (tsl's swap rule) inside the cluster's window -- in a per-thread slice of the LDS up to GRID_REPLAY_L
//      entries, in global memory beyond;
//   F  iteration order -> kept point indices.
// A cluster longer than GRID_MAX_CLUSTER raises `fail`: the caller takes the general path.  Same container order, bit for bit
// (tests/test_gpu_voxel.py runs both paths against the oracle's robin-map replay).
// All workgroups must be resident for the barriers to pass: 256 x 256 threads and 48 KB of LDS each are a fraction of the chip, and a
// workgroup that finds the chip busy is placed when the kernels in front of it retire (they do not wait for this one).  A barrier that
// is not passed within GRID_SPIN_LIMIT polls (seconds) raises `abort`: every workgroup leaves, the host sees info[5] == 0 and takes
// the general path.  (Round 5's first attempt let the LAST workgroup to finish phase A run B .. F alone: one compute unit's
// dependent-latency loops, 0.5 ms at 20 000 points against 0.23 for the launches -- profiles/r05_time_api_onelaunch.txt.)
constexpr int GRID_T = 256;
constexpr int GRID_MAX_WG = 256;
constexpr int GRID_REPLAY_L = 16;
constexpr int GRID_MAX_CLUSTER = 192;
constexpr int64_t GRID_MAX_N = 1 << 18;   // points (B <= 2^19 buckets)
constexpr unsigned GRID_SPIN_LIMIT = 1u << 22;
constexpr int GRID_NONE = 0x7fffffff;

struct GridRobinArgs {
    const double* pts;
    int64_t n, stride;
    double vs;
    unsigned mul_y;
    int B;                 // buckets of the reserved table (power of two)
    int* owner; int* tmin; int hsize; int* slot_of;   // first-point table
    unsigned* ctl;         // [0] barrier arrivals, [1] abort, [2] wrap: entry below which a bucket stays empty, [3] rotation, [4] max distance, [5] fail, [6] barrier rounds completed (all 0 before the launch)
    int* part;             // [5][GRID_MAX_WG] per-workgroup totals of the grid scans
    int* vfirst32;         // [n] first point of voxel v
    unsigned* vhash;       // [n]
    int* hist;             // [B]
    int* sorted_v;         // [n] voxels by home bucket
    int* key_s;            // [n] their (rotated) home buckets
    int* cm;               // [n] running maximum of home - index
    int* cl_start;         // [n + 1]
    int* tab_dist; int* tab_id;   // [n] the table, cluster windows back to back
    int64_t* keep_out;     // [n]
    int64_t* count_out;    // [1]
    int64_t* info;         // {B, nv (-1: a cluster beyond the limit / a wrap that could not be placed), max distance, wrapped} (0 before the launch)
    long long* trace;      // [32] wall-clock stamps (100 MHz) of workgroup 0 behind every barrier (vfm_debug_voxel_trace), or NULL
    // chained form (vfm_voxel_robin_level): point i is pts[idx[i]] (idx == NULL: pts[i]), moved by the 4 x 4 pose T first (NULL: as it is),
    // the number of points is read on the device (n_dev, <= n: the launch is sized for n), keep_out holds indices INTO pts (idx composed),
    // keep_local the positions in the level's own input
    const int64_t* idx;
    const int64_t* n_dev;
    const double* T;
    int64_t* keep_local;
};
// a level's point: gathered, then moved as vfm_transform_xyz_f64 moves it (csrc/project.hip dot4: the same operations in the same order)
__device__ __forceinline__ Vox grid_voxel_of(const GridRobinArgs& a, int i) {
    const double* p = a.pts + (a.idx ? a.idx[i] : (int64_t)i) * a.stride;
    if (!a.T) return voxel_of(p, a.vs);
    const double x = p[0], y = p[1], z = p[2];
    double q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) q[r] = ((a.T[4 * r] * x + a.T[4 * r + 1] * y) + a.T[4 * r + 2] * z) + a.T[4 * r + 3] * 1.0;
    return voxel_of(q, a.vs);
}

__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every workgroup of the grid arrives, then all leave; false: the grid gave up (abort flag), the caller returns at once.
// ctl[0] counts arrivals (one atomic per workgroup and barrier); the workgroup that completes a round publishes the round in ctl[6],
// which is what the others poll (plain agent-scope loads of a word nobody adds to).  The agent-scope release (L2 write-back) and
// acquire (L1 / L2 invalidate) are per compute unit / per L2, not per thread: thread 0 issues them for its workgroup, on either side
// of the workgroup's own barrier -- issued by all 65 536 threads they were most of the kernel's time.
__device__ __forceinline__ bool grid_sync(unsigned* ctl, unsigned target, int* sh_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's stores and atomics have reached the L2 (a workgroup-scope barrier alone need not wait for them)
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned prev = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (ordered by the fences around it)
        int ok = 1;
        if (prev + 1u == target) {
            __hip_atomic_store(ctl + 6, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (ld_agent(ctl + 6) < target) {   // (relaxed: an acquire here would invalidate the caches at every poll; the fence below does it once)
                if (ld_agent(ctl + 1) != 0u) { ok = 0; break; }
                if (++spins > GRID_SPIN_LIMIT) {
                    __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *sh_flag = ok;
    }
    __syncthreads();
    return *sh_flag != 0;
}
// 256 threads: sum / maximum over the threads in front of this one, and over the workgroup
__device__ __forceinline__ int block_excl_sum(int v, int* sh4, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) sh4[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < GRID_T / 64; ++w) {
        const int s = sh4[w];
        if (w < wave) base += s;
        tot += s;
    }
    total = tot;
    return base + incl - v;
}
__device__ __forceinline__ int block_excl_max(int v, int* sh4, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl = max(incl, t);
    }
    int excl = __shfl_up(incl, 1);
    if (lane == 0) excl = -GRID_NONE;
    __syncthreads();
    if (lane == 63) sh4[wave] = incl;
    __syncthreads();
    int tot = -GRID_NONE;
#pragma unroll
    for (int w = 0; w < GRID_T / 64; ++w) {
        const int s = sh4[w];
        if (w < wave) excl = max(excl, s);
        tot = max(tot, s);
    }
    total = tot;
    return excl;
}
// the totals of the workgroups in front of this one (sum / maximum), and of the grid
__device__ __forceinline__ int grid_prefix_sum(const int* part, int G, int* sh4, int* sh_b, int& total) {
    const int v = (int)threadIdx.x < G ? ld_agent(part + threadIdx.x) : 0;
    const int ex = block_excl_sum(v, sh4, total);
    if (threadIdx.x == blockIdx.x) *sh_b = ex;
    __syncthreads();
    return *sh_b;
}
__device__ __forceinline__ int grid_prefix_max(const int* part, int G, int* sh4, int* sh_b) {
    const int v = (int)threadIdx.x < G ? ld_agent(part + threadIdx.x) : -GRID_NONE;
    int total;
    const int ex = block_excl_max(v, sh4, total);
    if (threadIdx.x == blockIdx.x) *sh_b = ex;
    __syncthreads();
    return *sh_b;
}

__global__ __launch_bounds__(GRID_T) void voxel_robin_grid_kernel(GridRobinArgs a) {
    __shared__ int sh4[GRID_T / 64];
    __shared__ int sh_b, sh_c, sh_flag;
    __shared__ int sV[GRID_REPLAY_L][GRID_T], sI[GRID_REPLAY_L][GRID_T];        // [slot][thread]: conflict-free
    __shared__ short sH[GRID_REPLAY_L][GRID_T], sD[GRID_REPLAY_L][GRID_T];
    __shared__ int wv_mem[GRID_T / 64][GRID_MAX_CLUSTER], ws_mem[GRID_T / 64][GRID_MAX_CLUSTER], wi_mem[GRID_T / 64][GRID_MAX_CLUSTER];   // a wave's long cluster
    __shared__ short wh_mem[GRID_T / 64][GRID_MAX_CLUSTER], wd_mem[GRID_T / 64][GRID_MAX_CLUSTER];
    const int tid = threadIdx.x, G = (int)gridDim.x, T = G * GRID_T, gt = (int)blockIdx.x * GRID_T + tid;
    int n = (int)a.n, B = a.B;
    if (a.n_dev) {   // chained level: the previous level's count; reserve(n) -> B = 2^ceil(log2(ceil(float(n) / 0.5f))) as the host computes it
        const int64_t nd = *a.n_dev;
        n = (int)(nd < a.n ? nd : a.n);
        const float c = ceilf((float)n / 0.5f);
        const long long want = (long long)c;
        B = 1;
        while (B < want) B <<= 1;
    }
    const unsigned mask = (unsigned)(B - 1);
    unsigned arrivals = 0;
    int stamp = 0;
    if (a.trace && gt == 0) a.trace[stamp++] = wall_clock64();
#define GRID_SYNC()                                             \
    do {                                                        \
        arrivals += (unsigned)G;                                \
        if (!grid_sync(a.ctl, arrivals, &sh_flag)) return;      \
        if (a.trace && gt == 0 && stamp < 31) a.trace[stamp++] = wall_clock64();   \
    } while (0)
    // ---- 0: tables
    for (int s = gt; s < a.hsize; s += T) {
        a.owner[s] = EMPTY_OWNER;
        a.tmin[s] = 0x7fffffff;
    }
    for (int b = gt; b < B; b += T) a.hist[b] = 0;
    if (gt == 0) {
        __hip_atomic_store(a.ctl + 2, (unsigned)GRID_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.ctl + 3, (unsigned)GRID_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    GRID_SYNC();
    // ---- A: first-point table; a thread owns a run of consecutive points (one point up to 65 536)
    const int pc = (n + T - 1) / T;
    const int plo = min(n, gt * pc), phi = min(n, plo + pc);
    for (int i = plo; i < phi; ++i) {
        const Vox v = grid_voxel_of(a, i);
        int s = (int)(slot_hash(v) & (unsigned)(a.hsize - 1));
        while (true) {
            const int prev = atomicCAS(a.owner + s, EMPTY_OWNER, i);
            if (prev == EMPTY_OWNER) break;
            const Vox o = grid_voxel_of(a, prev);
            if (o.x == v.x && o.y == v.y && o.z == v.z) break;
            s = (s + 1) & (a.hsize - 1);
        }
        a.slot_of[i] = s;
        atomicMin(a.tmin + s, i);
    }
    GRID_SYNC();
    // ---- B: voxels in arrival order
    int nv;
    {
        int cnt = 0;
        for (int i = plo; i < phi; ++i) cnt += a.tmin[a.slot_of[i]] == i ? 1 : 0;
        int wg_total;
        const int ex = block_excl_sum(cnt, sh4, wg_total);
        if (tid == 0) st_agent(a.part + blockIdx.x, wg_total);
        GRID_SYNC();
        int pos = grid_prefix_sum(a.part, G, sh4, &sh_b, nv) + ex;
        for (int i = plo; i < phi; ++i)
            if (a.tmin[a.slot_of[i]] == i) {
                const unsigned h = reference_hash(grid_voxel_of(a, i), a.mul_y);
                a.vfirst32[pos] = i;
                a.vhash[pos] = h;
                atomicAdd(a.hist + (int)(h & mask), 1);
                ++pos;
            }
    }
    GRID_SYNC();
    const int vc = (nv + T - 1) / T, bc = (B + T - 1) / T;
    const int vlo = min(nv, gt * vc), vhi = min(nv, vlo + vc);
    const int blo = min(B, gt * bc), bhi = min(B, blo + bc);
    unsigned z = 0;
    int wrapped = 0, fail = 0, ncl = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {   // once more with the homes rotated by z
            for (int b = blo; b < bhi; ++b) a.hist[b] = 0;
            GRID_SYNC();
            for (int v = vlo; v < vhi; ++v) atomicAdd(a.hist + (int)((a.vhash[v] - z) & mask), 1);
            GRID_SYNC();
        }
        // ---- C: counting sort by home bucket
        {
            int sum = 0;
            for (int b = blo; b < bhi; ++b) sum += a.hist[b];
            int wg_total;
            const int ex = block_excl_sum(sum, sh4, wg_total);
            if (tid == 0) st_agent(a.part + GRID_MAX_WG + blockIdx.x, wg_total);
            GRID_SYNC();
            int all;
            int off = grid_prefix_sum(a.part + GRID_MAX_WG, G, sh4, &sh_b, all) + ex;
            for (int b = blo; b < bhi; ++b) {
                const int c = a.hist[b];
                a.hist[b] = off;
                off += c;
            }
        }
        GRID_SYNC();
        for (int v = vlo; v < vhi; ++v) {
            const int h = (int)((a.vhash[v] - z) & mask);
            const int p = atomicAdd(a.hist + h, 1);
            a.sorted_v[p] = v;
            a.key_s[p] = h;
        }
        GRID_SYNC();
        // ---- D: running maximum of d_i = home_i - i, clusters
        int run0, w;
        {
            int lmax = -GRID_NONE;
            for (int i = vlo; i < vhi; ++i) lmax = max(lmax, a.key_s[i] - i);
            int wg_max;
            const int ex = block_excl_max(lmax, sh4, wg_max);
            if (tid == 0) st_agent(a.part + 2 * GRID_MAX_WG + blockIdx.x, wg_max);
            GRID_SYNC();
            run0 = max(ex, grid_prefix_max(a.part + 2 * GRID_MAX_WG, G, sh4, &sh_b));   // over the entries in front of this thread's
            int run = run0, nflag = 0, nwrap = 0;
            for (int i = vlo; i < vhi; ++i) {
                const int d = a.key_s[i] - i;
                if (i == 0 || d > run) ++nflag;
                run = max(run, d);
                a.cm[i] = run;
                if (i + run >= B) ++nwrap;
            }
            int ftot, wtot;
            const int fex = block_excl_sum(nflag, sh4, ftot);
            (void)block_excl_sum(nwrap, sh4, wtot);
            if (tid == 0) {
                st_agent(a.part + 3 * GRID_MAX_WG + blockIdx.x, ftot);
                st_agent(a.part + 4 * GRID_MAX_WG + blockIdx.x, wtot);
            }
            GRID_SYNC();
            int cbase = grid_prefix_sum(a.part + 3 * GRID_MAX_WG, G, sh4, &sh_b, ncl) + fex;
            (void)grid_prefix_sum(a.part + 4 * GRID_MAX_WG, G, sh4, &sh_c, w);
            run = run0;
            for (int i = vlo; i < vhi; ++i) {
                const int d = a.key_s[i] - i;
                if (i == 0 || d > run) a.cl_start[cbase++] = i;
                run = max(run, d);
            }
            if (gt == 0) a.cl_start[ncl] = nv;
        }
        if (w > 0) {
            if (pass == 1) { fail = 1; break; }
            // smallest i with cm[i] >= w + 1: the bucket just below its cluster stays empty once the w wrapped entries have landed
            int cand = GRID_NONE;
            for (int i = vlo; i < vhi; ++i)
                if (a.cm[i] >= w + 1) { cand = i; break; }
            if (cand != GRID_NONE) atomicMin(a.ctl + 2, (unsigned)cand);
            GRID_SYNC();
            const int ibest = (int)ld_agent(a.ctl + 2);
            if (ibest == GRID_NONE) { fail = 1; break; }
            z = (unsigned)(ibest + a.cm[ibest] - 1);
            wrapped = 1;
            continue;
        }
        if (z != 0) {   // actual bucket = (rotated bucket + z) mod B: iteration starts at rotated bucket B - z
            int cand = GRID_NONE;
            for (int i = vlo; i < vhi; ++i)
                if (i + a.cm[i] >= B - (int)z) { cand = i; break; }
            if (cand != GRID_NONE) atomicMin(a.ctl + 3, (unsigned)cand);
        }
        break;
    }
    GRID_SYNC();
    // ---- E: one thread per cluster replays the container; a cluster beyond GRID_REPLAY_L entries is taken by the thread's whole wave
    //      (members ranked and their homes fetched by all lanes, the insertions by lane 0, everything in the LDS): replayed by its one
    //      thread in global memory such a cluster was a chain of ~L^2 / 4 dependent round trips -- 20 of the kernel's 59 us at 1 700
    //      points, 100 of 245 at 60 000, where the longest cluster of the half-empty table has 30 - 60 entries
    {
        int maxd = 0, myfail = 0;
        const int lane = tid & 63, wave = tid >> 6;
        int* wV = wv_mem[wave];
        int* wS = ws_mem[wave];
        int* wI = wi_mem[wave];
        short* wH = wh_mem[wave];
        short* wD = wd_mem[wave];
        if (!fail)
            for (int c0 = gt - lane; c0 < ncl; c0 += T) {   // (wave-uniform bounds: the lanes of a wave hold consecutive clusters)
                const int c = c0 + lane;
                const bool valid = c < ncl;
                const int i0 = valid ? a.cl_start[c] : 0, L = valid ? a.cl_start[c + 1] - i0 : 0;
                if (L > GRID_MAX_CLUSTER) myfail = 1;
                const int base = valid ? a.key_s[i0] : 0;
                if (valid && L <= GRID_REPLAY_L) {
                    // (an entry's home bucket lies beside it in key_s: fetched with the members, carried through their sort -- looked up
                    // from the voxel's hash behind the sort it was one more dependent round trip per cluster)
                    for (int t = 0; t < L; ++t) {
                        sV[t][tid] = a.sorted_v[i0 + t];
                        sH[t][tid] = (short)(a.key_s[i0 + t] - base);
                    }
                    for (int x = 1; x < L; ++x) {   // members by arrival (= voxel rank)
                        const int v = sV[x][tid];
                        const short hv = sH[x][tid];
                        int y = x - 1;
                        while (y >= 0 && sV[y][tid] > v) { sV[y + 1][tid] = sV[y][tid]; sH[y + 1][tid] = sH[y][tid]; --y; }
                        sV[y + 1][tid] = v;
                        sH[y + 1][tid] = hv;
                    }
                    for (int t = 0; t < L; ++t) sD[t][tid] = -1;
                    for (int t = 0; t < L; ++t) {
                        int v = sV[t][tid], ib = sH[t][tid], d = 0;
                        for (;;) {   // tsl insert_value_on_rehash (== insert_impl + insert_value_impl for an absent key)
                            const int rd = sD[ib][tid];
                            if (d > rd) {
                                if (rd < 0) { sD[ib][tid] = (short)d; sI[ib][tid] = v; if (d > maxd) maxd = d; break; }
                                const int tv = sI[ib][tid];
                                sD[ib][tid] = (short)d; sI[ib][tid] = v;
                                if (d > maxd) maxd = d;
                                d = rd; v = tv;
                            }
                            d++;
                            ib++;   // never leaves [0, L): the window is the cluster's final extent
                        }
                    }
                    for (int t = 0; t < L; ++t) a.tab_id[i0 + t] = sI[t][tid];
                }
                unsigned long long todo = __ballot(valid && L > GRID_REPLAY_L && L <= GRID_MAX_CLUSTER);
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                    const int ci0 = __shfl(i0, src), cL = __shfl(L, src), cbase = __shfl(base, src);
                    for (int t = lane; t < cL; t += 64) wV[t] = a.sorted_v[ci0 + t];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    for (int t = lane; t < cL; t += 64) {   // rank = members that arrived earlier (voxel ranks are distinct)
                        const int v = wV[t];
                        int r = 0;
                        for (int u = 0; u < cL; ++u) r += wV[u] < v ? 1 : 0;
                        wS[r] = v;
                        wH[r] = (short)((int)((a.vhash[v] - z) & mask) - cbase);
                        wD[t] = -1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (lane == 0)
                        for (int t = 0; t < cL; ++t) {
                            int v = wS[t], ib = wH[t], d = 0;
                            for (;;) {
                                const int rd = wD[ib];
                                if (d > rd) {
                                    if (rd < 0) { wD[ib] = (short)d; wI[ib] = v; if (d > maxd) maxd = d; break; }
                                    const int tv = wI[ib];
                                    wD[ib] = (short)d; wI[ib] = v;
                                    if (d > maxd) maxd = d;
                                    d = rd; v = tv;
                                }
                                d++;
                                ib++;
                            }
                        }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    for (int t = lane; t < cL; t += 64) a.tab_id[ci0 + t] = wI[t];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        if (maxd > 0) atomicMax(a.ctl + 4, (unsigned)maxd);
        if (myfail) atomicMax(a.ctl + 5, 1u);
    }
    GRID_SYNC();
    // ---- F: iteration order -> kept point indices
    if (!fail) fail = ld_agent(a.ctl + 5) != 0u;
    if (!fail) {
        int rot = 0;
        if (z != 0) {
            const unsigned r = ld_agent(a.ctl + 3);
            rot = r < (unsigned)nv ? (int)r : nv;
        }
        for (int i = gt; i < nv; i += T) {
            int j = i + rot;
            if (j >= nv) j -= nv;
            const int first = a.vfirst32[a.tab_id[j]];
            a.keep_out[i] = a.idx ? a.idx[first] : (int64_t)first;
            if (a.keep_local) a.keep_local[i] = (int64_t)first;
        }
    }
    if (gt == 0) {
        // (a level that was not reproduced publishes NO survivors -- keep_out was not written, and the next level of a chain
        //  reads its points through it: count 0 makes that level an empty one instead of a gather through uninitialised indices.  The
        //  launchers also clear count_out before the launch: a grid that gave up at a barrier never gets here.)
        *a.count_out = fail ? 0 : nv;
        a.info[0] = B;
        a.info[1] = fail ? -1 : nv;      // (0, as the host left it: the grid gave up at a barrier -- or an empty chained level: info[5] tells)
        a.info[5] = 1;
        a.info[2] = (int64_t)ld_agent(a.ctl + 4);
        a.info[3] = wrapped;
        if (a.trace) {
            a.trace[stamp++] = wall_clock64();
            a.trace[31] = stamp;
        }
    }
#undef GRID_SYNC
}

// How many workgroups of voxel_robin_grid_kernel can be resident at once on the compute units `st` may use (the grid-wide
// barriers need every workgroup of the launch resident; the grid was sized for 256 free compute units whatever the device or the
// stream): occupancy per compute unit (LDS: ~60 KB per workgroup) x the compute units in the stream's mask (hipExtStreamGetCUMask:
// a stream made by hipExtStreamCreateWithCUMask reports its own, any other the device's).  The kernel's loops are grid-stride, so a
// smaller grid still covers n.  0: the occupancy query failed -- the caller takes the general path.
static int grid_resident_limit(hipStream_t st) {
    static std::atomic<int> per_cu{-1};
    static std::atomic<int> dev_cus{0};
    int pc = per_cu.load(std::memory_order_relaxed);
    if (pc < 0) {
        int nb = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, voxel_robin_grid_kernel, GRID_T, 0) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        dev_cus.store(prop.multiProcessorCount, std::memory_order_relaxed);
        per_cu.store(nb, std::memory_order_relaxed);
        pc = nb;
    }
    int cus = dev_cus.load(std::memory_order_relaxed);
    uint32_t mask[16] = {0};
    if (hipExtStreamGetCUMask(st, 16, mask) == hipSuccess) {
        int bits = 0;
        for (int i = 0; i < 16; ++i) bits += __builtin_popcount(mask[i]);
        if (bits > 0 && bits < cus) cus = bits;
    } else {
        (void)hipGetLastError();
    }
    // one workgroup per compute unit is what the barrier cost was measured with; more than that only where the units are few
    long long lim = (long long)pc * cus;
    return (int)(lim < GRID_MAX_WG ? lim : GRID_MAX_WG);
}
// A grid that did not become resident (other tenants on the device: a second rank, a masked stream beside a long kernel) costs
// GRID_SPIN_LIMIT polls before it gives up.  After one such call the next GRID_BACKOFF calls of the same host thread go straight to the general path.
constexpr int GRID_BACKOFF = 64;
static thread_local int t_grid_backoff = 0;   // (thread-local, like the last-error string: the library keeps no process-global mutable state)

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
    int* hist_small;   // [2^ceil(log2 2n)] bucket histogram of voxel_robin_grid_kernel
    int64_t* gridctl;  // [8] info + [8] control words of voxel_robin_grid_kernel
    int* gridpart;     // [5][256] its per-workgroup totals
    long long* gridtrace;  // [32] vfm_debug_voxel_trace
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
        {
            size_t hb = 2;
            while (hb < 2 * nn) hb <<= 1;
            w.hist_small = c.take<int>(n <= GRID_MAX_N ? 2 * hb : 1);   // (B = the power of two >= 2 reserve_n, reserve_n = n)
            w.gridctl = c.take<int64_t>(16);
            w.gridpart = c.take<int>(5 * GRID_MAX_WG);
            w.gridtrace = c.take<long long>(32);
        }
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


// tools: the phase stamps of the last one-launch VoxelDownsample in `ws` (n as at that call), 100 MHz ticks; out_host[31] = their number
VFM_EXPORT int vfm_debug_voxel_trace(void* ws, int64_t n, int64_t* out_host) {
    VFM_CHECK_ARG(ws && out_host && n > 0, "voxel_trace: bad arguments");
    VoxelWs w = carve_voxel(ws, n, true);
    VFM_CHECK_HIP(hipDeviceSynchronize());
    VFM_CHECK_HIP(hipMemcpy(out_host, w.gridtrace, 32 * sizeof(long long), hipMemcpyDeviceToHost));
    return VFM_OK;
}

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

// One level of a CHAIN of VoxelDownsample()s (registration_node.py:399-414: .5 -> 1.0 -> 5.0 on the survivors), without a read-back: the
// level's points are pts[idx[i]] for i < *n_dev (idx == NULL: pts[i]; n_dev == NULL: n_max), moved by the pose T first if T != NULL
// (as vfm_transform_xyz_f64 moves them); keep_out receives the survivors in the container's order as indices INTO pts (what the next level
// takes as its idx), keep_local_out (nullable) their positions in this level's input, count_out their number, info_dev (device int64[8])
// {buckets, voxels or -1 = not reproduced by this kernel, largest probe distance, wrapped, -, 1 = ran to its end}.  Nothing is
// synchronised: a caller enqueues the levels of its chain and reads counts and infos once.  The one-launch kernel only (n_max <= 2^18);
// a level it cannot reproduce (a cluster beyond its limit: info[1] = -1, or a grid that did not become resident: info[5] = 0) is the
// caller's to redo through vfm_voxel_robin.  `ws`: vfm_voxel_robin_workspace_bytes(n_max), a workspace of its own per level in flight.
VFM_EXPORT int vfm_voxel_robin_level(const double* pts, int64_t stride, const int64_t* idx, int64_t n_max, const int64_t* n_dev,
                                     const double* T_dev, double voxel_size, uint32_t hash_mul_y, int64_t* keep_out,
                                     int64_t* keep_local_out, int64_t* count_out, int64_t* info_dev, void* ws, size_t ws_bytes,
                                     vfm_stream_t stream) {
    VFM_CHECK_ARG(pts && keep_out && count_out && info_dev && ws && stride >= 3, "voxel_robin_level: bad arguments");
    VFM_CHECK_ARG(voxel_size > 0.0 && n_max >= 1 && n_max <= GRID_MAX_N, "voxel_robin_level: bad voxel_size / n_max (1 .. 2^18)");
    if (ws_bytes < vfm_voxel_robin_workspace_bytes(n_max)) return vfm_fail(VFM_EWORKSPACE, "voxel_robin_level: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    VoxelWs w = carve_voxel(ws, n_max, true);
    int64_t B = 1;
    {
        const float c = ceilf((float)n_max / 0.5f);
        const int64_t want = (int64_t)c;
        while (B < want) B <<= 1;
    }
    GridRobinArgs a{};
    a.pts = pts; a.n = n_max; a.stride = stride; a.vs = voxel_size; a.mul_y = hash_mul_y; a.B = (int)B;
    a.owner = w.owner; a.tmin = w.tmin; a.hsize = (int)w.hsize; a.slot_of = w.slot_of;
    a.info = info_dev;
    a.ctl = reinterpret_cast<unsigned*>(w.gridctl + 8);
    a.part = w.gridpart;
    a.trace = nullptr;
    a.vfirst32 = w.seq_a; a.vhash = w.vhash; a.hist = w.hist_small;
    a.sorted_v = w.seq_b; a.key_s = reinterpret_cast<int*>(w.key_s); a.cm = w.cm; a.cl_start = w.cl_start;
    a.tab_dist = w.tab_dist; a.tab_id = w.tab_id; a.keep_out = keep_out; a.count_out = count_out;
    a.idx = idx; a.n_dev = n_dev; a.T = T_dev; a.keep_local = keep_local_out;
    VFM_CHECK_HIP(hipMemsetAsync(w.gridctl, 0, 16 * sizeof(int64_t), st));
    VFM_CHECK_HIP(hipMemsetAsync(info_dev, 0, 8 * sizeof(int64_t), st));
    VFM_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int64_t), st));   // a level that fails or gives up leaves an EMPTY level to the next one
    const int resident = grid_resident_limit(st);
    if (resident <= 0) return VFM_OK;   // (info stays 0: "not reproduced by this kernel", the caller's to redo through vfm_voxel_robin)
    const int ppt = vfm_cfg().voxel_grid_ppt > 0 ? vfm_cfg().voxel_grid_ppt : (n_max <= 32768 ? 1 : n_max <= 131072 ? 2 : 4);
    const int64_t per_wg = (int64_t)GRID_T * ppt;
    const int64_t want_wg = (n_max + per_wg - 1) / per_wg;
    const unsigned grid = (unsigned)(want_wg < resident ? want_wg : resident);
    hipLaunchKernelGGL(voxel_robin_grid_kernel, dim3(grid), dim3(GRID_T), 0, st, a);
    VFM_CHECK_LAUNCH("voxel_robin_grid_kernel");
    return VFM_OK;
}

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
    if (K == 1 && reserve_n >= 0 && n <= GRID_MAX_N && vfm_cfg().voxel_small) {
        // VoxelDownsample of a cloud of this size: one generation of a reserved table in ONE launch (voxel_robin_grid_kernel)
        int64_t B = 0;
        {
            const float c = ceilf((float)reserve_n / 0.5f);
            const int64_t want = (int64_t)c;
            if (want > 0) {
                B = 1;
                while (B < want) B <<= 1;
            }
        }
        const int resident = grid_resident_limit(st);
        bool backoff = false;
        if (t_grid_backoff > 0) {
            --t_grid_backoff;
            backoff = true;
        }
        if (B >= 2 * n && B <= (1ll << 20) && reserve_n == n && resident > 0 && !backoff) {   // (no rehash: nv <= n <= the load threshold B / 2; the histogram is sized for reserve(n))
            GridRobinArgs a{};
            a.pts = pts; a.n = n; a.stride = stride; a.vs = voxel_size; a.mul_y = hash_mul_y; a.B = (int)B;
            a.owner = w.owner; a.tmin = w.tmin; a.hsize = (int)w.hsize; a.slot_of = w.slot_of;
            a.info = w.gridctl;
            a.ctl = reinterpret_cast<unsigned*>(w.gridctl + 8);
            a.part = w.gridpart;
            a.trace = vfm_cfg().voxel_trace ? w.gridtrace : nullptr;
            a.vfirst32 = w.seq_a; a.vhash = w.vhash; a.hist = w.hist_small;
            a.sorted_v = w.seq_b; a.key_s = reinterpret_cast<int*>(w.key_s); a.cm = w.cm; a.cl_start = w.cl_start;
            a.tab_dist = w.tab_dist; a.tab_id = w.tab_id; a.keep_out = keep_out; a.count_out = count_out;
            VFM_CHECK_HIP(hipMemsetAsync(w.gridctl, 0, 16 * sizeof(int64_t), st));
            // (a barrier costs ~2.5 us + 0.04 us per workgroup, a second point per thread a dependent round trip in every phase:
            //  tools/ab_voxel_grid.py -- 60 000 points: 0.24 / 0.19 / 0.20 ms at 1 / 2 / 4 points per thread, 20 000: 0.130 / 0.125 / 0.140)
            const int ppt = vfm_cfg().voxel_grid_ppt > 0 ? vfm_cfg().voxel_grid_ppt : (n <= 32768 ? 1 : n <= 131072 ? 2 : 4);
            const int64_t per_wg = (int64_t)GRID_T * ppt;
            const int64_t want_wg = (n + per_wg - 1) / per_wg;
            const unsigned grid = (unsigned)(want_wg < resident ? want_wg : resident);
            hipLaunchKernelGGL(voxel_robin_grid_kernel, dim3(grid), dim3(GRID_T), 0, st, a);
            VFM_CHECK_LAUNCH("voxel_robin_grid_kernel");
            int64_t local[4];
            int64_t* gi = info_host ? info_host : local;   // (straight into the caller's array: a DMA without a staging copy if it is page-locked)
            VFM_CHECK_HIP(hipMemcpyAsync(gi, w.gridctl, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
            VFM_CHECK_HIP(hipStreamSynchronize(st));
            if (gi[1] > 0) return VFM_OK;
            if (gi[1] == 0) t_grid_backoff = GRID_BACKOFF;   // the grid gave up at a barrier: not resident
            if (info_host) info_host[0] = info_host[1] = info_host[2] = info_host[3] = 0;
            // (a cluster beyond the kernel's limit, a wrap it could not place, or a grid that did not become resident: the general path decides)
        }
    }
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
            if (vfm_cfg().voxel_replay2) {   // round 5: the replay orders a cluster's members itself and runs in the LDS (robin_replay2_kernel)
                hipLaunchKernelGGL(robin_replay2_kernel, dim3(blocks_of(m, 64)), dim3(64), 0, st, w.cl_start, w.cl_base, w.cidx, m, w.pos_s,
                                   cur, w.vhash, mask, z, w.tab_dist, w.tab_id, w.geninfo);
            } else {
                int cbits = 1;
                while ((1ll << cbits) < m) ++cbits;
                tb = w.cub_bytes;
                VFM_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(w.cub, tb, w.cl_of_pos, w.cl_sorted, cur, w.arrivals, (int)m, 0,
                                                                cbits, st));
                hipLaunchKernelGGL(robin_replay_kernel, dim3(blocks_of(m, 64)), dim3(64), 0, st, w.cl_start, w.cl_base, w.cidx, m,
                                   w.arrivals, w.vhash, mask, z, w.tab_dist, w.tab_id, w.geninfo);
            }
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
