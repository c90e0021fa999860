// icp.hip -- point-to-point ICP refinement on MI355X (gfx950), row F2 of the scope table.
//
// Replaces the device-side work of kiss_icp::RegisterFrame (Registration.cpp:145-195, called through
// register_frame, kiss_icp/registration.py:28-73, from registration_node.py:338-344):
//   icp_nearest_kernel   VoxelHashMap::GetCorrespondences (VoxelHashMap.cpp:76-168): nearest map point
//                        among the 27 voxels around each source point (<= 20 points per voxel),
//                        accepted if closer than max_correspondence_distance
//   icp_system_kernel    BuildLinearSystem (Registration.cpp:96-141): sum of J^T w J (6x6) and J^T w r
//                        with J = [I | -hat(s)], w = k^2 / (k + |r|^2)^2
// The 6x6 LDLT solve and SE3::exp (Registration.cpp:176-177) stay on the host (vfmreg/icp.py).
//
// The voxel grid is a sorted-key CSR (keys ascending, points of a voxel in insertion order), so the
// scan order over neighbours equals the reference's (voxel loops i, j, k ascending, then insertion
// order, strict '<' keeps the first minimum).  fp64, -ffp-contract=off, operation order spelled out
// as in oracle/vfm_oracle.c; the reduction is a fixed tree (thread t owns pairs i = t mod 256 in
// ascending order, then a stride-halving tree), which the oracle replays, so every iterate is
// bit-identical to the oracle's (TBB's reduction order in the reference is unspecified).
#include "common.h"

namespace {

__device__ __forceinline__ long long voxel_key(int vx, int vy, int vz) {
    return ((long long)(vx + (1 << 20)) << 42) | ((long long)(vy + (1 << 20)) << 21) | (long long)(vz + (1 << 20));
}

// step: the Gauss-Newton update of the previous iteration (Registration.cpp:178-179, "Equation (12)") applied in the same
// launch -- T by value, the arithmetic of vfm_transform_xyz_f64 -- and the moved points written to src_out
struct IcpStep {
    double T[12];
    int apply;
};
// 32 lanes per source point, lane l < 27 takes neighbour voxel l = 9 (a - kx + 1) + 3 (b - ky + 1) + (c - kz + 1) -- the
// reference's loop order -- : one binary search and at most a voxel's points per lane instead of 27 searches in a row (a
// point's chain of ~460 dependent loads was the whole 0.135 ms of an iteration at 20 000 points).  The lanes' candidates are
// merged by (distance, scan position): the first minimum of the reference's scan, the same squared distances.
constexpr int ICP_LANES = 32;
// (src and src_out may be the same array -- include/vfmreg.h; icp.py transforms in place from the second iteration on -- so
// neither is __restrict__; a group reads its point before lane 0 writes it, and no other group touches that point)
__global__ __launch_bounds__(256) void icp_nearest_kernel(const double* src, int64_t n,
                                                          const long long* __restrict__ keys,
                                                          const int* __restrict__ start,
                                                          const double* __restrict__ pts, int nv, double voxel_size,
                                                          double max_dist, double* __restrict__ tgt,
                                                          uint8_t* __restrict__ valid, IcpStep step,
                                                          double* src_out) {
    const int l = threadIdx.x & (ICP_LANES - 1);
    int64_t i = (int64_t)blockIdx.x * (256 / ICP_LANES) + (threadIdx.x / ICP_LANES);
    const bool live = i < n;
    // (the shuffles below want every lane of the wave: a group past the end runs on the origin and stores nothing)
    double px = 0.0, py = 0.0, pz = 0.0;
    if (live) {
        px = src[3 * i];
        py = src[3 * i + 1];
        pz = src[3 * i + 2];
    }
    if (step.apply) {
        const double* T = step.T;
        const double qx = ((T[0] * px + T[1] * py) + T[2] * pz) + T[3] * 1.0;
        const double qy = ((T[4] * px + T[5] * py) + T[6] * pz) + T[7] * 1.0;
        const double qz = ((T[8] * px + T[9] * py) + T[10] * pz) + T[11] * 1.0;
        px = qx; py = qy; pz = qz;
        if (l == 0 && live) {
            src_out[3 * i] = px;
            src_out[3 * i + 1] = py;
            src_out[3 * i + 2] = pz;
        }
    }
    const int kx = (int)(px / voxel_size), ky = (int)(py / voxel_size), kz = (int)(pz / voxel_size);
    double bx = 0.0, by = 0.0, bz = 0.0, best = 1.7976931348623157e308;
    unsigned order = 0xFFFFFFFFu;   // scan position of the lane's best point: (neighbour << 20) | index inside the voxel; none yet
    if (l < 27) {
        const long long key = voxel_key(kx - 1 + l / 9, ky - 1 + (l / 3) % 3, kz - 1 + l % 3);
        int lo = 0, hi = nv;  // lower_bound
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        if (lo < nv && keys[lo] == key) {
            const int j0 = start[lo], j1 = start[lo + 1];
            for (int j = j0; j < j1; ++j) {
                const double dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 < best) {
                    best = d2;
                    bx = pts[3 * j];
                    by = pts[3 * j + 1];
                    bz = pts[3 * j + 2];
                    order = ((unsigned)l << 20) | (unsigned)min(j - j0, (1 << 20) - 1);
                }
            }
        }
    }
#pragma unroll
    for (int off = ICP_LANES / 2; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(best, off), ox = __shfl_xor(bx, off), oy = __shfl_xor(by, off), oz = __shfl_xor(bz, off);
        const unsigned oo = __shfl_xor(order, off);
        // strict '<' in scan order: the smaller distance, and of equal distances the earlier position (a lane without a point
        // holds the largest position and never wins against one that has a point: its `best` is the initial value, which no
        // accepted point carries)
        const bool take = oo != 0xFFFFFFFFu && (order == 0xFFFFFFFFu || ob < best || (ob == best && oo < order));
        if (take) {
            best = ob; bx = ox; by = oy; bz = oz; order = oo;
        }
    }
    if (l != 0 || !live) return;
    // (closest - point).norm() < max_correspondence_distance (VoxelHashMap.cpp:147)
    const bool ok = order != 0xFFFFFFFFu && (sqrt(best) < max_dist);
    tgt[3 * i] = bx;
    tgt[3 * i + 1] = by;
    tgt[3 * i + 2] = bz;
    valid[i] = ok ? 1 : 0;
}

// ---- RegisterFrame(std::vector<Eigen::VectorXd>, ...) (Registration.cpp:384-423; round 6): the same loop on rows that carry descriptors of
// any width, with VoxelHashMap::GetCorrespondences(VectorXdVector) (VoxelHashMap.cpp:321-448) as its search: among the points of the 27
// voxels the FIRST minimum of   d = |xyz_n - xyz_p|^2 x c,   c = clamp(0.5 (1 - cos(desc_n, desc_p)), 0.01, 1)  if both descriptors
// have a non-zero element sum, else 1;  cos = dot / (|desc_n| |desc_p| + 1e-5);  accepted if the EUCLIDEAN distance is below
// max_correspondence_distance.  Sums in column order, no fused multiply-add (oracle/vfm_oracle.c orc_icp_nearest_desc replays them; Eigen's
// own reduction order is not specified -- the arg-min can differ from the reference's only between candidates whose d agree to the last bits).
__global__ __launch_bounds__(256) void icp_desc_stats_kernel(const double* __restrict__ desc, int64_t n, int f, double* __restrict__ norm_out,
                                                             uint8_t* __restrict__ has_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* r = desc + i * f;
    double ss = 0.0, sm = 0.0;
    for (int k = 0; k < f; ++k) {
        ss = ss + r[k] * r[k];
        sm = sm + r[k];
    }
    norm_out[i] = sqrt(ss);
    has_out[i] = sm != 0.0 ? 1 : 0;
}
struct IcpDesc {
    const double* src_desc;   // [n][f]
    const double* src_norm;   // [n]
    const uint8_t* src_has;   // [n]
    const double* map_desc;   // [m][f], rows in the grid's (CSR) order
    const double* map_norm;
    const uint8_t* map_has;
    int f;
};
__global__ __launch_bounds__(256) void icp_nearest_desc_kernel(const double* src, int64_t n, const long long* __restrict__ keys,
                                                               const int* __restrict__ start, const double* __restrict__ pts, int nv,
                                                               double voxel_size, double max_dist, double* __restrict__ tgt,
                                                               uint8_t* __restrict__ valid, IcpStep step, double* src_out, IcpDesc dd) {
    const int l = threadIdx.x & (ICP_LANES - 1);
    int64_t i = (int64_t)blockIdx.x * (256 / ICP_LANES) + (threadIdx.x / ICP_LANES);
    const bool live = i < n;
    double px = 0.0, py = 0.0, pz = 0.0;
    if (live) {
        px = src[3 * i];
        py = src[3 * i + 1];
        pz = src[3 * i + 2];
    }
    if (step.apply) {
        const double* T = step.T;
        const double qx = ((T[0] * px + T[1] * py) + T[2] * pz) + T[3] * 1.0;
        const double qy = ((T[4] * px + T[5] * py) + T[6] * pz) + T[7] * 1.0;
        const double qz = ((T[8] * px + T[9] * py) + T[10] * pz) + T[11] * 1.0;
        px = qx; py = qy; pz = qz;
        if (l == 0 && live) {
            src_out[3 * i] = px;
            src_out[3 * i + 1] = py;
            src_out[3 * i + 2] = pz;
        }
    }
    const int kx = (int)(px / voxel_size), ky = (int)(py / voxel_size), kz = (int)(pz / voxel_size);
    double bx = 0.0, by = 0.0, bz = 0.0, best = 1.7976931348623157e308;
    unsigned order = 0xFFFFFFFFu;
    if (l < 27 && live) {
        const long long key = voxel_key(kx - 1 + l / 9, ky - 1 + (l / 3) % 3, kz - 1 + l % 3);
        int lo = 0, hi = nv;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        if (lo < nv && keys[lo] == key) {
            const int j0 = start[lo], j1 = start[lo + 1];
            const double* pd = dd.src_desc + i * dd.f;
            const double pn = dd.src_norm[i];
            const bool phas = dd.f > 0 && dd.src_has[i] != 0;
            for (int j = j0; j < j1; ++j) {
                const double dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
                double d = (dx * dx + dy * dy) + dz * dz;
                if (dd.f > 0) {
                    double c = 1.0;
                    if (phas && dd.map_has[j] != 0) {
                        const double* nd = dd.map_desc + (int64_t)j * dd.f;
                        double dot = 0.0;
                        for (int k = 0; k < dd.f; ++k) dot = dot + nd[k] * pd[k];
                        const double cs = dot / (dd.map_norm[j] * pn + 1e-5);
                        c = 0.5 * (1.0 - cs);
                        c = c < 0.01 ? 0.01 : (1.0 < c ? 1.0 : c);   // std::clamp(c, 0.01, 1.0)
                    }
                    d = d * c;
                }
                if (d < best) {
                    best = d;
                    bx = pts[3 * j];
                    by = pts[3 * j + 1];
                    bz = pts[3 * j + 2];
                    order = ((unsigned)l << 20) | (unsigned)min(j - j0, (1 << 20) - 1);
                }
            }
        }
    }
#pragma unroll
    for (int off = ICP_LANES / 2; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(best, off), ox = __shfl_xor(bx, off), oy = __shfl_xor(by, off), oz = __shfl_xor(bz, off);
        const unsigned oo = __shfl_xor(order, off);
        const bool take = oo != 0xFFFFFFFFu && (order == 0xFFFFFFFFu || ob < best || (ob == best && oo < order));
        if (take) {
            best = ob; bx = ox; by = oy; bz = oz; order = oo;
        }
    }
    if (l != 0 || !live) return;
    // (closest_neighbor.head<3>() - point.head<3>()).norm() < max_correspondance_distance (VoxelHashMap.cpp:425-432): the Euclidean
    // distance of the chosen neighbour, not its weighted one
    const double ex = bx - px, ey = by - py, ez = bz - pz;
    const bool ok = order != 0xFFFFFFFFu && (sqrt((ex * ex + ey * ey) + ez * ez) < max_dist);
    tgt[3 * i] = bx;
    tgt[3 * i + 1] = by;
    tgt[3 * i + 2] = bz;
    valid[i] = ok ? 1 : 0;
}

// out[0..35] = J^T W J (row-major 6x6), out[36..41] = J^T W r, out[42] = number of pairs.
// One workgroup per output value: the 43 sums are independent chains, so workgroup k recomputes w and the two Jacobian columns it
// needs and adds ITS term of every pair in the order the oracle replays (thread t owns pairs i = t mod 256 ascending, then the
// stride-halving tree) -- the same operations on the same operands as one workgroup holding all 43 accumulators per thread
// (0.059 ms per iteration at 20 000 pairs: 79 pairs x 43 chains per thread on one compute unit), 43 times as wide.
__device__ __forceinline__ double icp_jacobian(int row, int col, const double* s) {   // J = [ I | -hat(s) ]
    if (col < 3) return row == col ? 1.0 : 0.0;
    const int c = col - 3;
    if (row == c) return 0.0;
    // -hat(s) = [[0, s2, -s1], [-s2, 0, s0], [s1, -s0, 0]]
    if (row == 0) return c == 1 ? s[2] : -s[1];
    if (row == 1) return c == 0 ? -s[2] : s[0];
    return c == 0 ? s[1] : -s[0];
}
__global__ __launch_bounds__(256) void icp_system_kernel(const double* __restrict__ src, const double* __restrict__ tgt,
                                                         const uint8_t* __restrict__ valid, int64_t n, double kernel,
                                                         double* __restrict__ out) {
    __shared__ double red[256];
    const int t = threadIdx.x, k = blockIdx.x;   // k: a * 6 + b (J^T W J), 36 + a (J^T W r), 42 (count)
    const int a = k < 36 ? k / 6 : k - 36, b = k % 6;
    double acc = 0.0;
    // eight pairs' loads in flight, their terms added in ascending order (a pair at a time had been a chain of 79 round trips)
    constexpr int U = 8;
    for (int64_t i0 = t; i0 < n; i0 += 256 * U) {
        double sv[U][3], tv[U][3];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + 256 * (int64_t)u;
            ok[u] = i < n && valid[i] != 0;
            const int64_t j = i < n ? i : i0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sv[u][c] = src[3 * j + c];
                tv[u][c] = tgt[3 * j + c];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            if (k == 42) {
                acc = acc + 1.0;
                continue;
            }
            const double* s = sv[u];
            const double r[3] = {s[0] - tv[u][0], s[1] - tv[u][1], s[2] - tv[u][2]};
            const double r2 = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
            const double w = (kernel * kernel) / ((kernel + r2) * (kernel + r2));
            const double jw[3] = {icp_jacobian(0, a, s) * w, icp_jacobian(1, a, s) * w, icp_jacobian(2, a, s) * w};
            if (k < 36)
                acc = acc + ((jw[0] * icp_jacobian(0, b, s) + jw[1] * icp_jacobian(1, b, s)) + jw[2] * icp_jacobian(2, b, s));
            else
                acc = acc + ((jw[0] * r[0] + jw[1] * r[1]) + jw[2] * r[2]);
        }
    }
    red[t] = acc;
    __syncthreads();
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if (t < stride) red[t] = red[t] + red[t + stride];
        __syncthreads();
    }
    if (t == 0) out[k] = red[0];
}

}  // namespace

VFM_EXPORT int vfm_icp_nearest(const double* src, int64_t n, const int64_t* keys, const int32_t* start, const double* pts,
                               int32_t n_voxels, double voxel_size, double max_dist, double* tgt_out, uint8_t* valid_out,
                               vfm_stream_t stream) {
    VFM_CHECK_ARG(src && keys && start && pts && tgt_out && valid_out && n >= 0 && n_voxels >= 0 && voxel_size > 0.0,
                  "icp_nearest: bad arguments");
    if (n == 0) return VFM_OK;
    IcpStep step;
    step.apply = 0;
    hipLaunchKernelGGL(icp_nearest_kernel, dim3((unsigned)((n + 256 / ICP_LANES - 1) / (256 / ICP_LANES))), dim3(256), 0, (hipStream_t)stream, src, n,
                       reinterpret_cast<const long long*>(keys), start, pts, n_voxels, voxel_size, max_dist, tgt_out,
                       valid_out, step, (double*)nullptr);
    VFM_CHECK_LAUNCH("icp_nearest_kernel");
    return VFM_OK;
}

VFM_EXPORT int vfm_icp_step_nearest(const double* src, int64_t n, const double* T_host, double* src_out, const int64_t* keys,
                                    const int32_t* start, const double* pts, int32_t n_voxels, double voxel_size, double max_dist,
                                    double* tgt_out, uint8_t* valid_out, vfm_stream_t stream) {
    VFM_CHECK_ARG(src && T_host && src_out && keys && start && pts && tgt_out && valid_out && n >= 0 && n_voxels >= 0 && voxel_size > 0.0,
                  "icp_step_nearest: bad arguments");
    if (n == 0) return VFM_OK;
    IcpStep step;
    for (int k = 0; k < 12; ++k) step.T[k] = T_host[k];
    step.apply = 1;
    hipLaunchKernelGGL(icp_nearest_kernel, dim3((unsigned)((n + 256 / ICP_LANES - 1) / (256 / ICP_LANES))), dim3(256), 0, (hipStream_t)stream, src, n,
                       reinterpret_cast<const long long*>(keys), start, pts, n_voxels, voxel_size, max_dist, tgt_out,
                       valid_out, step, src_out);
    VFM_CHECK_LAUNCH("icp_nearest_kernel(step)");
    return VFM_OK;
}

VFM_EXPORT int vfm_icp_desc_stats(const double* desc, int64_t n, int32_t f, double* norm_out, uint8_t* has_out, vfm_stream_t stream) {
    VFM_CHECK_ARG(desc && norm_out && has_out && n >= 0 && f >= 1, "icp_desc_stats: bad arguments");
    if (n == 0) return VFM_OK;
    hipLaunchKernelGGL(icp_desc_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, desc, n, (int)f, norm_out, has_out);
    VFM_CHECK_LAUNCH("icp_desc_stats_kernel");
    return VFM_OK;
}

VFM_EXPORT int vfm_icp_step_nearest_desc(const double* src, int64_t n, const double* T_host, double* src_out, const double* src_desc,
                                         const double* src_norm, const uint8_t* src_has, int32_t f, const int64_t* keys, const int32_t* start,
                                         const double* pts, const double* map_desc, const double* map_norm, const uint8_t* map_has,
                                         int32_t n_voxels, double voxel_size, double max_dist, double* tgt_out, uint8_t* valid_out,
                                         vfm_stream_t stream) {
    VFM_CHECK_ARG(src && keys && start && pts && tgt_out && valid_out && n >= 0 && n_voxels >= 0 && voxel_size > 0.0 && f >= 1 && src_desc &&
                      src_norm && src_has && map_desc && map_norm && map_has && (!T_host || src_out),
                  "icp_step_nearest_desc: bad arguments");
    if (n == 0) return VFM_OK;
    IcpStep step;
    step.apply = T_host ? 1 : 0;
    for (int k = 0; k < 12; ++k) step.T[k] = T_host ? T_host[k] : 0.0;
    IcpDesc dd{src_desc, src_norm, src_has, map_desc, map_norm, map_has, (int)f};
    hipLaunchKernelGGL(icp_nearest_desc_kernel, dim3((unsigned)((n + 256 / ICP_LANES - 1) / (256 / ICP_LANES))), dim3(256), 0, (hipStream_t)stream,
                       src, n, reinterpret_cast<const long long*>(keys), start, pts, n_voxels, voxel_size, max_dist, tgt_out, valid_out, step,
                       src_out, dd);
    VFM_CHECK_LAUNCH("icp_nearest_desc_kernel");
    return VFM_OK;
}

VFM_EXPORT int vfm_icp_build_system(const double* src, const double* tgt, const uint8_t* valid, int64_t n, double kernel,
                                    double* out43, vfm_stream_t stream) {
    VFM_CHECK_ARG(src && tgt && valid && out43 && n >= 0, "icp_build_system: bad arguments");
    hipLaunchKernelGGL(icp_system_kernel, dim3(43), dim3(256), 0, (hipStream_t)stream, src, tgt, valid, n, kernel, out43);
    VFM_CHECK_LAUNCH("icp_system_kernel");
    return VFM_OK;
}
