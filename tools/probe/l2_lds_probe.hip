// What a compute unit of gfx950 can pull into its LDS by LDS-DMA (global_load_lds_dwordx4) as a function of the bytes it keeps in flight:
// every wave keeps D loads of 1 KiB outstanding (counted s_waitcnt vmcnt), W workgroups of 4 waves per compute unit (set by the LDS a
// workgroup asks for), source either a 1 MiB region every workgroup walks (L2-resident: the weight operand of a GEMM) or a region of
// its own per workgroup (streamed through the L2 from the Infinity Cache / HBM: the token operand).  Round 5: the ViT GEMMs sit at
// 5 - 8 TB/s of L2 -> LDS traffic whatever their tile shape; this is the ceiling they sit under.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_lds_probe tools/probe/l2_lds_probe.hip && /tmp/l2_lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
}

template <int D, bool TOREG>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, unsigned span_mask, unsigned own_bytes, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)wave * (unsigned)(D * 1024);
    // shared region: every wave starts at its own KiB and walks on by the grid's width; own region: the workgroup's slice
    const char* base = src + (size_t)blockIdx.x * own_bytes;
    unsigned pos = ((unsigned)(blockIdx.x * 4 + wave) * 1024u) & span_mask;
    const unsigned step = own_bytes ? 4096u : 1024u * 4u * 37u;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep)::"memory");
    uint4 r[D];
    auto issue = [&](int d) {
        const unsigned voff = pos + (unsigned)lane * 16u;
        if constexpr (TOREG) {
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[d]) : "v"(voff), "s"(base) : "memory");
        } else {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)d * 1024u);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "memory");
        }
        pos = (pos + step) & span_mask;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d);
    unsigned sink = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            wait_vmcnt<D - 1>();
            if constexpr (TOREG) sink += r[d].x;
            issue(d);
        }
    }
    wait_vmcnt<0>();
    asm volatile("s_mov_b32 m0, %0" ::"s"(keep) : "memory");
    if (sink == 0x12345u) out[0] = 1.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)lds[0];
}

template <int D, bool TOREG>
static void run(const char* src, bool shared, int wg_per_cu, float* out) {
    const int cus = 256;
    const int grid = cus * wg_per_cu;
    // LDS per workgroup so that exactly wg_per_cu fit a compute unit (160 KiB), at least the ring itself
    size_t lds = (size_t)(160 * 1024 / wg_per_cu) & ~(size_t)1023;
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<D, TOREG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (lds < (size_t)4 * D * 1024) { printf("  (D = %d does not fit %d workgroups per unit)\n", D, wg_per_cu); return; }
    const unsigned span_mask = shared ? (1u << 20) - 1u : (1u << 17) - 1u;   // (own region: half of the workgroup's 256 KiB, so that a wave's KiB never leaves it)
    const unsigned own = shared ? 0u : (1u << 18);
    const int iters = 2048 / D;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<D, TOREG>), dim3(grid), dim3(256), lds, 0, src, span_mask, own, 16, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<D, TOREG>), dim3(grid), dim3(256), lds, 0, src, span_mask, own, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)grid * 4.0 * (double)(iters * D + D) * 1024.0;
    printf("  %s, %2d waves per unit x %2d KiB in flight each = %4d KiB per unit: %7.2f TB/s (%6.1f GB/s per unit), %.3f ms\n",
           TOREG ? "registers" : "LDS-DMA  ", 4 * wg_per_cu, D, 4 * wg_per_cu * D, bytes / ms * 1e-9, bytes / ms * 1e-6 / cus, ms);
    fflush(stdout);
}

int main() {
    char* src;
    float* out;
    const size_t bytes = (size_t)2048 << 18;   // 2048 workgroups x 256 KiB
    CHECK(hipMalloc(&src, bytes + (1 << 16)));   // (a wave's last KiB may start at the region's last byte)
    CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMalloc(&out, 64));
    for (int shared = 1; shared >= 0; --shared) {
        printf("%s\n", shared ? "source: one 1 MiB region read by every workgroup (L2-resident)" : "source: 256 KiB of its own per workgroup, walked again and again (L2 / Infinity Cache)");
        for (int w : {1, 2, 3, 4, 8}) {
            run<1, false>(src, shared, w, out);
            run<2, false>(src, shared, w, out);
            run<4, false>(src, shared, w, out);
            run<8, false>(src, shared, w, out);
            run<16, false>(src, shared, w, out);
        }
        if (shared)
        for (int w : {1, 3, 8}) {
            run<2, true>(src, shared, w, out);
            run<8, true>(src, shared, w, out);
            run<16, true>(src, shared, w, out);
        }
    }
    return 0;
}
