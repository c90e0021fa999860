// Do LDS reads cost MFMA time?  Per wave and iteration: 8 independent v_mfma_f32_32x32x16_f16 (4 accumulators) and R ds_read_b128 of
// 1 KiB each (conflict-free, results never consumed by the MFMAs), R = 0 .. 16 -- the LDS-tiled ViT GEMM reads 8 per 8 MFMAs, the
// token-stationary one 4 -- at 1 .. 4 waves per SIMD.  Also with LDS-DMA fills (global_load_lds_dwordx4 from an L2-resident region)
// at the GEMM's ratio.  Round 5, DESIGN.md R5.9: a stage of those kernels takes about the SUM of its LDS traffic and its MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_lds_probe tools/probe/mfma_lds_probe.hip && /tmp/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int R, int DMA>
__global__ void mix(float* out, const char* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i)); }
    floatx16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)wave * 8192u;
    const unsigned addr = base + (unsigned)lane * 16u;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep)::"memory");
    unsigned voff = ((unsigned)(blockIdx.x * 8 + wave) * 4096u + (unsigned)lane * 16u) & ((1u << 20) - 1u);
    for (int it = 0; it < iters; ++it) {
        intx4 r[R > 0 ? R : 1];
#pragma unroll
        for (int k = 0; k < R; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k]) : "v"(addr), "n"((k % 8) * 1024));
#pragma unroll
        for (int k = 0; k < DMA; ++k) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)k * 1024u);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory");
            voff = (voff + 65536u) & ((1u << 20) - 1u);
        }
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < R; ++k) asm volatile("" ::"v"(r[k]));
    }
    asm volatile("s_mov_b32 m0, %0" ::"s"(keep) : "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

template <int R, int DMA>
static void run(float* out, const char* src, int waves_per_simd) {
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;   // one workgroup per compute unit
    const int wgs = 256 * (256 * waves_per_simd / threads);
    const size_t lds = (size_t)(threads / 64) * 8192;
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mix<R, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    hipLaunchKernelGGL((mix<R, DMA>), dim3(wgs), dim3(threads), lds, 0, out, src, 50);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix<R, DMA>), dim3(wgs), dim3(threads), lds, 0, out, src, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)wgs * (threads / 64);
    const double flops = waves * iters * 8.0 * 2.0 * 32 * 32 * 16;
    printf("  %d waves per SIMD, %2d LDS reads + %d LDS-DMA KiB per 8 MFMAs: %7.1f TFLOP/s; LDS reads %6.1f TB/s, fills %5.1f TB/s\n", waves_per_simd, R, DMA,
           flops / ms * 1e-9, waves * iters * R * 1024.0 / ms * 1e-9, waves * iters * DMA * 1024.0 / ms * 1e-9);
    fflush(stdout);
}

int main() {
    float* out;
    char* src;
    CHECK(hipMalloc(&out, 4096 * 1024 * 4));
    CHECK(hipMalloc(&src, (1 << 20) + 65536));
    CHECK(hipMemset(src, 1, (1 << 20) + 65536));
    for (int w : {1, 2, 3, 4}) {
        run<0, 0>(out, src, w);
        run<4, 0>(out, src, w);
        run<8, 0>(out, src, w);
        run<16, 0>(out, src, w);
        run<0, 4>(out, src, w);
        run<8, 4>(out, src, w);
        run<4, 2>(out, src, w);
    }
    return 0;
}
