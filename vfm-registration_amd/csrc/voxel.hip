// voxel.hip -- voxel down-sampling / voxel-hash-map insertion on MI355X (gfx950), row F1.
//
// Replaces kiss_icp::VoxelDownsample (Preprocessing.cpp:50-137: first point of every voxel) and the
// selection rule of VoxelHashMap::AddPoints (VoxelHashMap.cpp:733-770 + VoxelHashMap.hpp:55-62: at
// most max_points_per_voxel points per voxel, in insertion order):
//     keep point i  <=>  fewer than K earlier points (j < i) fall into the same voxel,
//     voxel = trunc(xyz / voxel_size) per axis (Eigen cast<int>, Preprocessing.cpp:58).
// The reference walks the cloud sequentially through a tsl::robin_map; here the same set is computed
// in parallel and exactly: an open-addressing hash table maps voxel keys to slots (atomicCAS), then
// K rounds of "smallest not-yet-taken index per crowded voxel" (atomicMin) pick the first K points of
// the voxels that hold more than K.  Survivors are emitted in input order (the reference: hash-map
// iteration order -- same set).  HBM-bound integer work: coalesced SoA passes, atomics
// only on the (L2-resident) table.
#include "common.h"

namespace {

constexpr long long EMPTY_KEY = -1;

__device__ __forceinline__ long long voxel_key_of(const double* __restrict__ p, double vs) {
    const int vx = (int)(p[0] / vs), vy = (int)(p[1] / vs), vz = (int)(p[2] / vs);
    return ((long long)(vx + (1 << 20)) << 42) | ((long long)(vy + (1 << 20)) << 21) | (long long)(vz + (1 << 20));
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {  // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

__global__ __launch_bounds__(256) void voxel_init_kernel(long long* __restrict__ tkeys, int* __restrict__ tcount,
                                                         int* __restrict__ tmin, int64_t hsize) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= hsize) return;
    tkeys[s] = EMPTY_KEY;
    tcount[s] = 0;
    tmin[s] = 0x7fffffff;
}

__global__ __launch_bounds__(256) void voxel_insert_kernel(const double* __restrict__ pts, int64_t n, int64_t stride,
                                                           double vs, long long* __restrict__ tkeys,
                                                           int* __restrict__ tcount, int* __restrict__ tmin,
                                                           int64_t hmask, int* __restrict__ slot_of) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long key = voxel_key_of(pts + i * stride, vs);
    int64_t s = (int64_t)(mix64((unsigned long long)key) & (unsigned long long)hmask);
    while (true) {
        const long long prev = (long long)atomicCAS(reinterpret_cast<unsigned long long*>(tkeys + s),
                                                    (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (prev == EMPTY_KEY || prev == key) break;
        s = (s + 1) & hmask;
    }
    slot_of[i] = (int)s;
    atomicAdd(tcount + s, 1);
    atomicMin(tmin + s, (int)i);
}

// state: 1 keep, 0 drop, 2 undecided (crowded voxel, K > 1)
__global__ __launch_bounds__(256) void voxel_classify_kernel(int64_t n, int K, const int* __restrict__ tcount,
                                                             const int* __restrict__ tmin,
                                                             const int* __restrict__ slot_of, uint8_t* __restrict__ state) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int s = slot_of[i];
    uint8_t st;
    if (tcount[s] <= K) st = 1;
    else if (tmin[s] == (int)i) st = 1;       // the first point of a crowded voxel is always kept
    else st = (K == 1) ? 0 : 2;
    state[i] = st;
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

// stable compaction of the kept indices by one workgroup (ascending = input order)
__global__ __launch_bounds__(1024) void voxel_compact_kernel(const uint8_t* __restrict__ state, int64_t n,
                                                             int64_t* __restrict__ keep, int64_t* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int64_t base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int64_t s = 0; s < n; s += 1024) {
        const int64_t i = s + threadIdx.x;
        const bool valid = (i < n) && state[i] == 1;
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
        if (valid) keep[base + woff + before] = i;
        __syncthreads();
        if (threadIdx.x == 0) base_s = base + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base_s;
}

struct VoxelWs {
    long long* tkeys;
    int* tcount;
    int* tmin;
    int* slot_of;
    uint8_t* state;
    int64_t hsize;
    size_t bytes;
};

inline VoxelWs carve_voxel(void* p, int64_t n) {
    VfmCarver c(p);
    VoxelWs w;
    int64_t h = 1024;
    while (h < 2 * n) h <<= 1;
    w.hsize = h;
    w.tkeys = c.take<long long>((size_t)h);
    w.tcount = c.take<int>((size_t)h);
    w.tmin = c.take<int>((size_t)h);
    w.slot_of = c.take<int>((size_t)(n > 0 ? n : 1));
    w.state = c.take<uint8_t>((size_t)(n > 0 ? n : 1));
    w.bytes = c.used();
    return w;
}

}  // namespace

VFM_EXPORT size_t vfm_voxel_first_workspace_bytes(int64_t n) { return carve_voxel(nullptr, n).bytes; }

VFM_EXPORT int vfm_voxel_first(const double* pts, int64_t n, int64_t stride, double voxel_size, int32_t max_per_voxel,
                               int64_t* keep_out, int64_t* count_out, void* ws, size_t ws_bytes, vfm_stream_t stream) {
    VFM_CHECK_ARG(pts && keep_out && count_out && ws && n >= 0 && stride >= 3, "voxel_first: bad arguments");
    VFM_CHECK_ARG(voxel_size > 0.0 && max_per_voxel >= 1 && n < (1ll << 31), "voxel_first: bad voxel_size / cap / n");
    if (ws_bytes < vfm_voxel_first_workspace_bytes(n)) return vfm_fail(VFM_EWORKSPACE, "voxel_first: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    VoxelWs w = carve_voxel(ws, n);
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(voxel_init_kernel, dim3((unsigned)((w.hsize + 255) / 256)), dim3(256), 0, st, w.tkeys, w.tcount, w.tmin,
                       w.hsize);
    if (n > 0) {
        hipLaunchKernelGGL(voxel_insert_kernel, dim3(gb), dim3(256), 0, st, pts, n, stride, voxel_size, w.tkeys, w.tcount,
                           w.tmin, w.hsize - 1, w.slot_of);
        hipLaunchKernelGGL(voxel_classify_kernel, dim3(gb), dim3(256), 0, st, n, (int)max_per_voxel, w.tcount, w.tmin,
                           w.slot_of, w.state);
        for (int r = 1; r < max_per_voxel; ++r) {  // rounds 2..K: next smallest undecided index per crowded voxel
            hipLaunchKernelGGL(voxel_round_reset_kernel, dim3(gb), dim3(256), 0, st, n, w.slot_of, w.state, w.tmin);
            hipLaunchKernelGGL(voxel_round_min_kernel, dim3(gb), dim3(256), 0, st, n, w.slot_of, w.state, w.tmin);
            hipLaunchKernelGGL(voxel_round_take_kernel, dim3(gb), dim3(256), 0, st, n, w.slot_of, w.state, w.tmin);
        }
    }
    hipLaunchKernelGGL(voxel_compact_kernel, dim3(1), dim3(1024), 0, st, w.state, n, keep_out, count_out);
    VFM_CHECK_LAUNCH("voxel_first kernels");
    return VFM_OK;
}
