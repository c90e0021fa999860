// Bare issue rate of v_mfma_f32_32x32x16_f16 (the ViT GEMMs' instruction) on gfx950: four independent accumulators per wave, constant
// operands, W waves per SIMD -- what the matrix pipes sustain inside the power envelope, to price the ViT kernels against
// (the spec's 2.5 PFLOP/s = one such MFMA per 32 clocks and SIMD at 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f16_mfma_probe tools/probe/f16_mfma_probe.hip && /tmp/f16_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void rate(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i)); }
    floatx16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 4096 * 1024 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int threads : {256, 512, 1024})
        for (int wgs : {256, 512, 1024}) {
            hipLaunchKernelGGL(rate, dim3(wgs), dim3(threads), 0, 0, out, 100);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(rate, dim3(wgs), dim3(threads), 0, 0, out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = (double)wgs * (threads / 64) * (double)iters * 4.0 * 2.0 * 32 * 32 * 16;
            const double per_simd = (double)wgs * (threads / 64) / 1024.0;   // waves per SIMD over the launch
            printf("%4d threads x %4d workgroups (%.2f waves per SIMD): %7.1f TFLOP/s, %.1f ns per MFMA and SIMD\n", threads, wgs, per_simd,
                   flops / ms * 1e-9, ms * 1e6 / ((double)iters * 4.0 * (per_simd < 1 ? 1 : per_simd)));
        }
    return 0;
}
