// What one MI355X delivers to kernels shaped like the operand preparation (csrc/match_prep.hip): 338 MB of fp32 rows read once.
//   stream:  grid-stride float4 loads, many small workgroups (the usual streaming form), U loads in flight per thread
//   group :  one workgroup of 512 threads per compute unit walks 196 KB groups: all of a group's loads are issued, then waited for
//            (prep_chunk_kernel's shape without its prefetch); with `ahead` the next group's loads are issued before the wait
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_probe tools/probe/hbm_probe.hip && /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ x, size_t n4, float* out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) {
                const float* p = reinterpret_cast<const float*>(x + i + u * stride);
                v[u].x = __builtin_nontemporal_load(p);
                v[u].y = __builtin_nontemporal_load(p + 1);
                v[u].z = __builtin_nontemporal_load(p + 2);
                v[u].w = __builtin_nontemporal_load(p + 3);
            } else {
                v[u] = x[i + u * stride];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) acc += x[i].x;
    if (acc == 12345.678f) out[0] = acc;
}

// rows of 384 floats; a group = 128 rows; wave w of 8 owns rows 16 w ..; lane l reads float4 l and (l < 32) float4 64 + l of a row
template <bool AHEAD>
__global__ __launch_bounds__(512) void group_kernel(const float* __restrict__ x, int groups, float* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 v[16][2], w[AHEAD ? 16 : 1][2];
    float acc = 0.f;
    auto load = [&](float4 (*dst)[2], int g) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float* row = x + ((size_t)g * 128 + wave * 16 + j) * 384;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane + 64 * i < 96) {
                    const float* p = row + 4 * (lane + 64 * i);
                    t.x = __builtin_nontemporal_load(p);
                    t.y = __builtin_nontemporal_load(p + 1);
                    t.z = __builtin_nontemporal_load(p + 2);
                    t.w = __builtin_nontemporal_load(p + 3);
                }
                dst[j][i] = t;
            }
        }
    };
    if ((int)blockIdx.x < groups) load(v, blockIdx.x);
    for (int g = blockIdx.x; g < groups; g += gridDim.x) {
        const int gn = g + gridDim.x;
        if (AHEAD && gn < groups) load(w, gn);
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc += v[j][i].x * v[j][i].y + v[j][i].z * v[j][i].w;
        if (AHEAD) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) v[j][i] = w[j][i];
        } else if (gn < groups) {
            load(v, gn);
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// read rows, write a quarter of the bytes (the int8 image's share) -- the mix of the preparation
template <int U>
__global__ __launch_bounds__(256) void mix_kernel(const float4* __restrict__ x, size_t n4, unsigned* __restrict__ y) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float* p = reinterpret_cast<const float*>(x + i + u * stride);
            v[u].x = __builtin_nontemporal_load(p);
            v[u].y = __builtin_nontemporal_load(p + 1);
            v[u].z = __builtin_nontemporal_load(p + 2);
            v[u].w = __builtin_nontemporal_load(p + 3);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned b = ((unsigned)(int)v[u].x & 255u) | (((unsigned)(int)v[u].y & 255u) << 8) | (((unsigned)(int)v[u].z & 255u) << 16) | ((unsigned)(int)v[u].w << 24);
            __builtin_nontemporal_store(b, y + i + u * stride);
        }
    }
}

template <typename F>
static float time_ms(F f, int reps = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int groups = 1720;   // 220 160 rows of 384 floats = 338 MB
    const size_t nfl = (size_t)groups * 128 * 384, n4 = nfl / 4;
    float *x, *out;
    unsigned* y;
    hipMalloc(&x, nfl * 4);
    hipMalloc(&y, n4 * 4);
    hipMalloc(&out, 64);
    hipMemset(x, 0, nfl * 4);
    const double gb = nfl * 4 / 1e9;
    for (int wg : {1024, 2048, 4096, 8192, 16384}) {
        float t1 = time_ms([&] { hipLaunchKernelGGL((stream_kernel<4, true>), dim3(wg), dim3(256), 0, 0, (const float4*)x, n4, out); });
        float t2 = time_ms([&] { hipLaunchKernelGGL((stream_kernel<8, true>), dim3(wg), dim3(256), 0, 0, (const float4*)x, n4, out); });
        float t3 = time_ms([&] { hipLaunchKernelGGL((stream_kernel<8, false>), dim3(wg), dim3(256), 0, 0, (const float4*)x, n4, out); });
        printf("stream  %5d workgroups x 256: 4 in flight nt %.3f ms = %.2f TB/s | 8 in flight nt %.3f ms = %.2f TB/s | 8 in flight cached %.3f ms = %.2f TB/s\n", wg, t1,
               gb / t1, t2, gb / t2, t3, gb / t3);
    }
    for (int wg : {256, 512}) {
        float t1 = time_ms([&] { hipLaunchKernelGGL((group_kernel<false>), dim3(wg), dim3(512), 0, 0, x, groups, out); });
        float t2 = time_ms([&] { hipLaunchKernelGGL((group_kernel<true>), dim3(wg), dim3(512), 0, 0, x, groups, out); });
        printf("group   %5d workgroups x 512: issue-wait %.3f ms = %.2f TB/s | next group ahead %.3f ms = %.2f TB/s\n", wg, t1, gb / t1, t2, gb / t2);
    }
    for (int wg : {2048, 8192}) {
        float t1 = time_ms([&] { hipLaunchKernelGGL((mix_kernel<8>), dim3(wg), dim3(256), 0, 0, (const float4*)x, n4, y); });
        printf("mix     %5d workgroups x 256: read 338 MB + write 85 MB %.3f ms = %.2f TB/s\n", wg, t1, gb * 1.25 / t1);
    }
    return 0;
}
