// project.hip -- LiDAR -> image projection and descriptor lifting on MI355X (gfx950).
//
//   vfm_project_pinhole_f64        Dataset.project_pcl_to_image (dataloader/nclt.py:311-366,
//                                  oxford_robotcar.py:330-363, kitti_odometry.py:110-125)
//   vfm_gather_bilinear_patchgrid  image_features.py:104-110 (bilinear upsample) fused with
//                                  prepare_scenes.py:57-62, 80-104 (black-pixel zeroing, NCLT
//                                  rot90, per-point gather, first-camera-wins scatter)
//   vfm_lift_multicam              the two above for all cameras of a scan in ONE launch (projection
//                                  fused with the gather; no compaction, no per-camera round trip)
//   vfm_transform_xyz_f64          vfm_reg/utils.py:47-54
//
// HBM-bound integer / fp64 work: one thread per point, coalesced SoA reads, a single-workgroup
// stable compaction (N is a LiDAR scan, 1e4..1e5 points).  fp64 / fp32 expressions are spelled
// in the oracle's order and the file is compiled with -ffp-contract=off, so pixel coordinates
// and surviving point indices are bit-identical to the reference's on the golden fixtures.
#include "common.h"

#include <string.h>
#include "../../include/vfmreg.h"

namespace {

__device__ __forceinline__ double dot4(const double* __restrict__ r, const double p[4]) {
    return ((r[0] * p[0] + r[1] * p[1]) + r[2] * p[2]) + r[3] * p[3];
}
__device__ __forceinline__ double dot3(const double* __restrict__ r, const double p[3]) {
    return (r[0] * p[0] + r[1] * p[1]) + r[2] * p[2];
}

struct ProjArgs {
    double mats[48];
    double fc[4];
    double subsample;
    long long win[4];
    long long H, W;
    int mode;
};

// projection of ONE point (homogeneous p) by one camera: the reference's three project_pcl_to_image
// variants, fp64, oracle operation order.  Returns "point survives"; (ui, vi) = integer pixel.
__device__ __forceinline__ bool project_one(const ProjArgs& a, const uint8_t* __restrict__ image, const double p[4],
                                            long long& ui, long long& vi) {
    const double* M0 = a.mats;
    const double* M1 = a.mats + 16;
    const double* M2 = a.mats + 32;
    ui = 0;
    vi = 0;
    bool keep = false;
    if (a.mode == VFM_PROJ_NCLT) {
        double pc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) pc[r] = dot4(M0 + 4 * r, p);
        const double q0 = dot3(M1 + 0, pc), q1 = dot3(M1 + 3, pc), q2 = dot3(M1 + 6, pc);
        const double x = q0 / q2 / a.subsample;
        const double y = q1 / q2 / a.subsample;
        if (q2 > 0.0 && x > -2147483648.0 && x < 2147483648.0 && y > -2147483648.0 && y < 2147483648.0) {
            ui = (long long)x;  // astype(int): truncation toward zero
            vi = (long long)y;
            if (ui >= a.win[1] && ui < a.win[1] + a.win[3] && vi >= a.win[0] && vi < a.win[0] + a.win[2]) {
                ui -= a.win[1];
                vi -= a.win[0];
                keep = true;
                if (image) {
                    const uint8_t* px = image + (vi * a.W + ui) * 3;
                    if (px[0] == 0 && px[1] == 0 && px[2] == 0) keep = false;
                }
            }
        }
    } else if (a.mode == VFM_PROJ_ROBOTCAR) {
        double e[4], c[4], g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = dot4(M0 + 4 * r, p);
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = dot4(M1 + 4 * r, e);
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = dot4(M2 + 4 * r, c);
        if (g[2] >= 0.0) {
            double uu = a.fc[0] * g[0] / g[2] + a.fc[2];
            double vv = a.fc[1] * g[1] / g[2] + a.fc[3];
            uu = uu / a.subsample;
            vv = vv / a.subsample;
            // inclusive upper bound: reference quirk (oxford_robotcar.py:356-357); NaN dropped
            if (uu >= 0.0 && uu <= (double)a.W && vv >= 0.0 && vv <= (double)a.H) {
                ui = (long long)uu;
                vi = (long long)vv;
                keep = true;
            }
        }
    } else {
        const double q0 = dot4(M0 + 0, p), q1 = dot4(M0 + 4, p), q2 = dot4(M0 + 8, p);
        if (q2 > 0.0) {
            const double uu = q0 / q2 / a.subsample;
            const double vv = q1 / q2 / a.subsample;
            if (uu >= 0.0 && uu <= (double)a.W && vv >= 0.0 && vv <= (double)a.H) {
                ui = (long long)uu;
                vi = (long long)vv;
                keep = true;
            }
        }
    }
    return keep;
}

__global__ __launch_bounds__(256) void project_points_kernel(const double* __restrict__ pcl, int64_t n, ProjArgs a,
                                                             const uint8_t* __restrict__ image,
                                                             int32_t* __restrict__ ucand, int32_t* __restrict__ vcand,
                                                             uint8_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double p[4] = {pcl[i], pcl[n + i], pcl[2 * n + i], pcl[3 * n + i]};
    long long ui, vi;
    const bool keep = project_one(a, image, p, ui, vi);
    ucand[i] = (int32_t)ui;
    vcand[i] = (int32_t)vi;
    flag[i] = keep ? 1 : 0;
}

// stable compaction by one workgroup: survivors keep ascending point order (np.where order)
__global__ __launch_bounds__(1024) void project_compact_kernel(const int32_t* __restrict__ ucand,
                                                               const int32_t* __restrict__ vcand,
                                                               const uint8_t* __restrict__ flag, int64_t n,
                                                               int32_t* __restrict__ u_out, int32_t* __restrict__ v_out,
                                                               int64_t* __restrict__ idx_out, int64_t* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int64_t base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int64_t s = 0; s < n; s += 1024) {
        const int64_t i = s + threadIdx.x;
        const bool valid = (i < n) && flag[i];
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
        if (valid) {
            const int64_t k = base + woff + before;
            u_out[k] = ucand[i];
            v_out[k] = vcand[i];
            idx_out[k] = i;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s = base + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base_s;
}

// torch upsample_bilinear2d(align_corners=False) source index / weights, fp32
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    int a = (int)src;
    if (a > in_size - 1) a = in_size - 1;
    const int b = a + ((a < in_size - 1) ? 1 : 0);
    float lam = src - (float)a;
    if (lam < 0.0f) lam = 0.0f;
    if (lam > 1.0f) lam = 1.0f;
    i0 = a;
    i1 = b;
    l1 = lam;
    l0 = 1.0f - lam;
}

// descriptor of ONE projected point (pixel u, v of the image the projection addressed), written by one
// wavefront: bilinear sample of the patch grid at the pixel (image_features.py:104-108 without the
// full-resolution tensor), zero if the raw image is black there (prepare_scenes.py:57-62).
// lanes sub = 0 .. nsub - 1 share the row (nsub = 64: one wavefront per point; 16: four points per wavefront).  ZERO: a point
// whose pixel is out of range or black gets its zeros written here (the caller need not clear the row beforehand).
template <int NSUB = 64, bool ZERO = false>
__device__ __forceinline__ void gather_point(const float* __restrict__ grid, int gh, int gw, int C, int Hup, int Wup,
                                             int rot_mode, const uint8_t* __restrict__ image, int u, int v,
                                             float* __restrict__ o, int lane) {
    int row, col;
    if (rot_mode == 1) {
        row = u;
        col = Wup - 1 - v;
    } else {
        row = v;
        col = u;
    }
    // The RobotCar / KITTI projections keep the reference's inclusive bound (u == W or v == H can be emitted,
    // oxford_robotcar.py:356-357); the reference's `feat[v, u, :]` raises IndexError there.  Here such a point keeps a
    // zero descriptor instead of reading past the image / the patch grid.
    bool black = row < 0 || row >= Hup || col < 0 || col >= Wup;
    if (!black && image) {
        const uint8_t* px = image + ((int64_t)row * Wup + col) * 3;
        black = (px[0] == 0 && px[1] == 0 && px[2] == 0);
    }
    if (black) {  // the descriptor is (stays) zero
        if constexpr (ZERO) {
            if ((C & 3) == 0)
                for (int c4 = lane; c4 < (C >> 2); c4 += NSUB) reinterpret_cast<float4*>(o)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
            else
                for (int c = lane; c < C; c += NSUB) o[c] = 0.0f;
        }
        return;
    }
    const float sh = (float)gh / (float)Hup;
    const float sw = (float)gw / (float)Wup;
    int h0, h1, w0, w1;
    float hl0, hl1, wl0, wl1;
    src_index(sh, row, gh, h0, h1, hl0, hl1);
    src_index(sw, col, gw, w0, w1, wl0, wl1);
    const float* f00 = grid + ((int64_t)h0 * gw + w0) * C;
    const float* f01 = grid + ((int64_t)h0 * gw + w1) * C;
    const float* f10 = grid + ((int64_t)h1 * gw + w0) * C;
    const float* f11 = grid + ((int64_t)h1 * gw + w1) * C;
    if ((C & 3) == 0) {
        for (int c4 = lane; c4 < (C >> 2); c4 += NSUB) {
            const float4 a = reinterpret_cast<const float4*>(f00)[c4];
            const float4 b = reinterpret_cast<const float4*>(f01)[c4];
            const float4 c = reinterpret_cast<const float4*>(f10)[c4];
            const float4 d = reinterpret_cast<const float4*>(f11)[c4];
            float4 r;
            r.x = hl0 * (wl0 * a.x + wl1 * b.x) + hl1 * (wl0 * c.x + wl1 * d.x);
            r.y = hl0 * (wl0 * a.y + wl1 * b.y) + hl1 * (wl0 * c.y + wl1 * d.y);
            r.z = hl0 * (wl0 * a.z + wl1 * b.z) + hl1 * (wl0 * c.z + wl1 * d.z);
            r.w = hl0 * (wl0 * a.w + wl1 * b.w) + hl1 * (wl0 * c.w + wl1 * d.w);
            reinterpret_cast<float4*>(o)[c4] = r;
        }
    } else {
        for (int c = lane; c < C; c += NSUB)
            o[c] = hl0 * (wl0 * f00[c] + wl1 * f01[c]) + hl1 * (wl0 * f10[c] + wl1 * f11[c]);
    }
}

// one wavefront per projected point of ONE camera; cameras are launched in priority order
__global__ __launch_bounds__(256) void gather_bilinear_kernel(const float* __restrict__ grid, int gh, int gw, int C,
                                                              int Hup, int Wup, int rot_mode,
                                                              const uint8_t* __restrict__ image,
                                                              const int32_t* __restrict__ u, const int32_t* __restrict__ v,
                                                              const int64_t* __restrict__ idx,
                                                              const int64_t* __restrict__ count_dev, int64_t k_max,
                                                              float* __restrict__ desc, uint8_t* __restrict__ filled) {
    const int64_t K = count_dev ? min(*count_dev, k_max) : k_max;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= K) return;
    const int lane = threadIdx.x & 63;
    const int64_t pt = idx[i];
    if (filled[pt]) return;  // an earlier (higher-priority) camera already owns this point
    gather_point(grid, gh, gw, C, Hup, Wup, rot_mode, image, u[i], v[i], desc + pt * (int64_t)C, lane);
    // black pixel: descriptor stays zero but the point is still claimed (prepare_scenes.py:57-62
    // zeroes the feature, np.unique at :96-101 still keeps this camera's entry)
    if (lane == 0) filled[pt] = 1;
}

// create_descriptors (prepare_scenes.py:50-107) for ALL cameras in one launch -- the projection fused with
// the gather.  One wavefront per LiDAR point: lane c projects the point into camera c (same device code as
// project_points_kernel), the first surviving camera in priority order wins (np.unique(return_index=True)
// at :96-101 keeps the first camera's entry), and the whole wavefront samples that camera's patch grid.
// Points seen by no camera keep their (zero-initialised) descriptor; `filled` reports which were seen.
constexpr int LIFT_MAX_CAMS = 6;  // kernel arguments are limited to 4 KiB
struct LiftCam {
    ProjArgs proj;
    const uint8_t* proj_image;
    const float* grid;
    const uint8_t* raw_image;
    int gh, gw, Hup, Wup, rot_mode;
};
struct LiftArgs {
    LiftCam cam[LIFT_MAX_CAMS];
    int ncam;
};

// Round 3: FOUR points per wavefront.  A point's work is a chain of dependent memory round trips (coordinates -> camera
// parameters -> pixel -> four grid rows -> store) that one wavefront per point ran once per 1.5 KB of output: 20 000 waves,
// 0.07 of the HBM rate.  Now lanes 0 .. 23 project (point j = lane / 6, camera lane % 6), the first surviving camera of each
// point comes out of one ballot, and lane group g = lane >> 4 samples point g's grid with 16 lanes (6 float4 per row and lane);
// the four chains of a wave run side by side.  Rows of points no camera sees (or whose pixel is black) are written as zeros
// here: desc_out no longer has to be cleared by the caller.
__global__ __launch_bounds__(256) void lift_multicam_kernel(const double* __restrict__ pcl, int64_t n, LiftArgs a, int C,
                                                            float* __restrict__ desc, uint8_t* __restrict__ filled) {
    const int lane = threadIdx.x & 63;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;   // the wave's first point
    if (i0 >= n) return;
    long long ui = 0, vi = 0;
    bool keep = false;
    {
        const int j = lane / 6, cam = lane - 6 * j;
        const int64_t i = i0 + j;
        if (lane < 24 && cam < a.ncam && i < n) {
            const double p[4] = {pcl[i], pcl[n + i], pcl[2 * n + i], pcl[3 * n + i]};
            keep = project_one(a.cam[cam].proj, a.cam[cam].proj_image, p, ui, vi);
        }
    }
    const unsigned long long seen_all = __ballot(keep);
    const int g = lane >> 4, sub = lane & 15;
    const int64_t i = i0 + g;
    const unsigned seen = (unsigned)((seen_all >> (6 * g)) & 63ull);
    // (every lane takes part in the shuffles; the source lane of a point without a camera is arbitrary)
    const int c = seen ? __builtin_ctz(seen) : 0;   // first camera in priority order
    const int u = __shfl((int)ui, 6 * g + c), v = __shfl((int)vi, 6 * g + c);
    if (i >= n) return;
    float* o = desc + i * (int64_t)C;
    if (seen == 0u) {
        if ((C & 3) == 0)
            for (int c4 = sub; c4 < (C >> 2); c4 += 16) reinterpret_cast<float4*>(o)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
        else
            for (int cc = sub; cc < C; cc += 16) o[cc] = 0.0f;
        if (sub == 0) filled[i] = 0;
        return;
    }
    // the four points of a wave may have chosen different cameras: lane-indexed camera record
    const LiftCam& cam = a.cam[c];
    gather_point<16, true>(cam.grid, cam.gh, cam.gw, C, cam.Hup, cam.Wup, cam.rot_mode, cam.raw_image, u, v, o, sub);
    if (sub == 0) filled[i] = 1;
}

__global__ __launch_bounds__(256) void transform_xyz_kernel(const double* __restrict__ xyz, int64_t n,
                                                            const double* __restrict__ T, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double p[4] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 1.0};
#pragma unroll
    for (int r = 0; r < 3; ++r) out[3 * i + r] = dot4(T + 4 * r, p);
}

}  // namespace

VFM_EXPORT size_t vfm_project_workspace_bytes(int64_t n) {
    VfmCarver c(nullptr);
    c.take<int32_t>((size_t)n);
    c.take<int32_t>((size_t)n);
    c.take<uint8_t>((size_t)n);
    return c.used();
}

VFM_EXPORT int vfm_project_pinhole_f64(int mode, const double* pcl, int64_t n, const double* mats_host,
                                       const double* fc_host, double subsample, const int64_t* win_host,
                                       const uint8_t* image, int64_t H, int64_t W, int32_t* u_out, int32_t* v_out,
                                       int64_t* idx_out, int64_t* count_out, void* ws, size_t ws_bytes,
                                       vfm_stream_t stream) {
    VFM_CHECK_ARG(mode >= 0 && mode <= 2, "project: unknown mode %d", mode);
    VFM_CHECK_ARG(pcl && mats_host && u_out && v_out && idx_out && count_out && n >= 0, "project: bad arguments");
    VFM_CHECK_ARG(subsample > 0.0, "project: subsample must be positive");
    if (ws_bytes < vfm_project_workspace_bytes(n)) return vfm_fail(VFM_EWORKSPACE, "project: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    ProjArgs a;
    for (int k = 0; k < 48; ++k) a.mats[k] = mats_host[k];
    for (int k = 0; k < 4; ++k) a.fc[k] = fc_host ? fc_host[k] : 0.0;
    for (int k = 0; k < 4; ++k) a.win[k] = win_host ? win_host[k] : 0;
    a.subsample = subsample;
    a.H = H;
    a.W = W;
    a.mode = mode;
    VfmCarver c(ws);
    int32_t* ucand = c.take<int32_t>((size_t)n);
    int32_t* vcand = c.take<int32_t>((size_t)n);
    uint8_t* flag = c.take<uint8_t>((size_t)n);
    if (n > 0) {
        hipLaunchKernelGGL(project_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pcl, n, a, image,
                           ucand, vcand, flag);
        VFM_CHECK_LAUNCH("project_points_kernel");
    }
    hipLaunchKernelGGL(project_compact_kernel, dim3(1), dim3(1024), 0, st, ucand, vcand, flag, n, u_out, v_out, idx_out,
                       count_out);
    VFM_CHECK_LAUNCH("project_compact_kernel");
    return VFM_OK;
}

VFM_EXPORT int vfm_gather_bilinear_patchgrid(const float* grid, int gh, int gw, int C, int Hup, int Wup, int rot_mode,
                                             const uint8_t* image, const int32_t* u, const int32_t* v,
                                             const int64_t* idx, const int64_t* count_dev, int64_t k_max,
                                             float* desc_out, uint8_t* filled, vfm_stream_t stream) {
    VFM_CHECK_ARG(grid && u && v && idx && desc_out && filled, "gather: null pointer");
    VFM_CHECK_ARG(gh > 0 && gw > 0 && C > 0 && Hup > 0 && Wup > 0 && k_max >= 0, "gather: bad sizes");
    if (k_max == 0) return VFM_OK;
    hipLaunchKernelGGL(gather_bilinear_kernel, dim3((unsigned)((k_max + 3) / 4)), dim3(256), 0, (hipStream_t)stream, grid,
                       gh, gw, C, Hup, Wup, rot_mode, image, u, v, idx, count_dev, k_max, desc_out, filled);
    VFM_CHECK_LAUNCH("gather_bilinear_kernel");
    return VFM_OK;
}

VFM_EXPORT int vfm_transform_xyz_f64(const double* xyz, int64_t n, const double* T, double* out, vfm_stream_t stream) {
    VFM_CHECK_ARG(xyz && T && out && n >= 0, "transform: bad arguments");
    if (n == 0) return VFM_OK;
    hipLaunchKernelGGL(transform_xyz_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, n,
                       T, out);
    VFM_CHECK_LAUNCH("transform_xyz_kernel");
    return VFM_OK;
}

VFM_EXPORT int vfm_lift_multicam(const double* pcl, int64_t n, int ncam, const vfm_lift_camera* cams_host, int C,
                                 float* desc_out, uint8_t* filled, vfm_stream_t stream) {
    VFM_CHECK_ARG(pcl && cams_host && desc_out && filled && n >= 0 && C > 0, "lift: bad arguments");
    VFM_CHECK_ARG(ncam >= 1 && ncam <= LIFT_MAX_CAMS, "lift: 1..%d cameras per call (got %d)", LIFT_MAX_CAMS, ncam);
    LiftArgs a;
    memset(&a, 0, sizeof(a));
    a.ncam = ncam;
    for (int c = 0; c < ncam; ++c) {
        const vfm_lift_camera& h = cams_host[c];
        VFM_CHECK_ARG(h.mode >= 0 && h.mode <= 2 && h.subsample > 0.0 && h.grid, "lift: camera %d: bad mode / subsample / grid", c);
        VFM_CHECK_ARG(h.gh > 0 && h.gw > 0 && h.Hup > 0 && h.Wup > 0, "lift: camera %d: bad sizes", c);
        LiftCam& d = a.cam[c];
        for (int k = 0; k < 48; ++k) d.proj.mats[k] = h.mats[k];
        for (int k = 0; k < 4; ++k) d.proj.fc[k] = h.fc[k];
        for (int k = 0; k < 4; ++k) d.proj.win[k] = h.win[k];
        d.proj.subsample = h.subsample;
        d.proj.H = h.H;
        d.proj.W = h.W;
        d.proj.mode = h.mode;
        d.proj_image = h.proj_image;
        d.grid = h.grid;
        d.raw_image = h.raw_image;
        d.gh = h.gh; d.gw = h.gw; d.Hup = h.Hup; d.Wup = h.Wup; d.rot_mode = h.rot_mode;
    }
    if (n == 0) return VFM_OK;
    hipLaunchKernelGGL(lift_multicam_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, pcl, n, a, C,
                       desc_out, filled);
    VFM_CHECK_LAUNCH("lift_multicam_kernel");
    return VFM_OK;
}
