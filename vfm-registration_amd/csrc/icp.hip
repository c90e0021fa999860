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
__global__ __launch_bounds__(256) void icp_nearest_kernel(const double* __restrict__ src, int64_t n,
                                                          const long long* __restrict__ keys,
                                                          const int* __restrict__ start,
                                                          const double* __restrict__ pts, int nv, double voxel_size,
                                                          double max_dist, double* __restrict__ tgt,
                                                          uint8_t* __restrict__ valid, IcpStep step,
                                                          double* __restrict__ src_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double px = src[3 * i], py = src[3 * i + 1], pz = src[3 * i + 2];
    if (step.apply) {
        const double* T = step.T;
        const double qx = ((T[0] * px + T[1] * py) + T[2] * pz) + T[3] * 1.0;
        const double qy = ((T[4] * px + T[5] * py) + T[6] * pz) + T[7] * 1.0;
        const double qz = ((T[8] * px + T[9] * py) + T[10] * pz) + T[11] * 1.0;
        px = qx; py = qy; pz = qz;
        src_out[3 * i] = px;
        src_out[3 * i + 1] = py;
        src_out[3 * i + 2] = pz;
    }
    const int kx = (int)(px / voxel_size), ky = (int)(py / voxel_size), kz = (int)(pz / voxel_size);
    double bx = 0.0, by = 0.0, bz = 0.0, best = 1.7976931348623157e308;
    bool found = false;
    for (int a = kx - 1; a <= kx + 1; ++a)
        for (int b = ky - 1; b <= ky + 1; ++b)
            for (int c = kz - 1; c <= kz + 1; ++c) {
                const long long key = voxel_key(a, b, c);
                int lo = 0, hi = nv;  // lower_bound
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (keys[mid] < key) lo = mid + 1; else hi = mid;
                }
                if (lo < nv && keys[lo] == key) {
                    for (int j = start[lo]; j < start[lo + 1]; ++j) {
                        const double dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
                        const double d2 = (dx * dx + dy * dy) + dz * dz;
                        if (d2 < best) {
                            best = d2;
                            bx = pts[3 * j];
                            by = pts[3 * j + 1];
                            bz = pts[3 * j + 2];
                            found = true;
                        }
                    }
                }
            }
    // (closest - point).norm() < max_correspondence_distance (VoxelHashMap.cpp:147)
    const bool ok = found && (sqrt(best) < max_dist);
    tgt[3 * i] = bx;
    tgt[3 * i + 1] = by;
    tgt[3 * i + 2] = bz;
    valid[i] = ok ? 1 : 0;
}

// out[0..35] = J^T W J (row-major 6x6), out[36..41] = J^T W r, out[42] = number of pairs
__global__ __launch_bounds__(256) void icp_system_kernel(const double* __restrict__ src, const double* __restrict__ tgt,
                                                         const uint8_t* __restrict__ valid, int64_t n, double kernel,
                                                         double* __restrict__ out) {
    __shared__ double red[43][256];
    const int t = threadIdx.x;
    double acc[43];
#pragma unroll
    for (int k = 0; k < 43; ++k) acc[k] = 0.0;
    for (int64_t i = t; i < n; i += 256) {
        if (!valid[i]) continue;
        const double s[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]};
        const double r[3] = {s[0] - tgt[3 * i], s[1] - tgt[3 * i + 1], s[2] - tgt[3 * i + 2]};
        const double r2 = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
        const double w = (kernel * kernel) / ((kernel + r2) * (kernel + r2));
        // J = [ I | -hat(s) ]
        const double J[3][6] = {{1.0, 0.0, 0.0, 0.0, s[2], -s[1]},
                                {0.0, 1.0, 0.0, -s[2], 0.0, s[0]},
                                {0.0, 0.0, 1.0, s[1], -s[0], 0.0}};
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double jw[3] = {J[0][a] * w, J[1][a] * w, J[2][a] * w};
#pragma unroll
            for (int b = 0; b < 6; ++b)
                acc[a * 6 + b] = acc[a * 6 + b] + ((jw[0] * J[0][b] + jw[1] * J[1][b]) + jw[2] * J[2][b]);
            acc[36 + a] = acc[36 + a] + ((jw[0] * r[0] + jw[1] * r[1]) + jw[2] * r[2]);
        }
        acc[42] = acc[42] + 1.0;
    }
#pragma unroll
    for (int k = 0; k < 43; ++k) red[k][t] = acc[k];
    __syncthreads();
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if (t < stride)
            for (int k = 0; k < 43; ++k) red[k][t] = red[k][t] + red[k][t + stride];
        __syncthreads();
    }
    if (t < 43) out[t] = red[t][0];
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
    hipLaunchKernelGGL(icp_nearest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, n,
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
    hipLaunchKernelGGL(icp_nearest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, n,
                       reinterpret_cast<const long long*>(keys), start, pts, n_voxels, voxel_size, max_dist, tgt_out,
                       valid_out, step, src_out);
    VFM_CHECK_LAUNCH("icp_nearest_kernel(step)");
    return VFM_OK;
}

VFM_EXPORT int vfm_icp_build_system(const double* src, const double* tgt, const uint8_t* valid, int64_t n, double kernel,
                                    double* out43, vfm_stream_t stream) {
    VFM_CHECK_ARG(src && tgt && valid && out43 && n >= 0, "icp_build_system: bad arguments");
    hipLaunchKernelGGL(icp_system_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src, tgt, valid, n, kernel, out43);
    VFM_CHECK_LAUNCH("icp_system_kernel");
    return VFM_OK;
}
