// How many workgroups of a given shape (threads, LDS bytes, VGPRs) run AT THE SAME TIME on this chip?  Every workgroup notes its start
// and end time (s_memrealtime) and the hardware id of its first wave, then spins ~100 us.  A grid of G workgroups that fits in one round
// ends at ~100 us; the first size that does not shows up as a second round.   hipcc --offload-arch=gfx950 -O3 -o /tmp/wgcap wg_capacity_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#include <algorithm>

#define SPIN(NAME, THREADS, VGPRS) \
__global__ __launch_bounds__(THREADS, 1) __attribute__((amdgpu_num_vgpr(VGPRS))) void NAME(unsigned long long* out, int spin_ticks) { \
    extern __shared__ unsigned char lds[]; \
    const unsigned long long t0 = __builtin_readcyclecounter(); \
    const unsigned long long r0 = wall_clock64(); \
    if (threadIdx.x == 0) lds[0] = 1; \
    __syncthreads(); \
    while ((long long)(wall_clock64() - r0) < spin_ticks) __builtin_amdgcn_s_sleep(8); \
    __syncthreads(); \
    if (threadIdx.x == 0) { \
        unsigned hw; \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); \
        unsigned xcc; \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
        out[blockIdx.x * 4 + 0] = r0; \
        out[blockIdx.x * 4 + 1] = wall_clock64(); \
        out[blockIdx.x * 4 + 2] = ((unsigned long long)(xcc & 0xf) << 32) | hw; \
        out[blockIdx.x * 4 + 3] = t0 + lds[0]; \
    } \
}
SPIN(spin_768_168, 768, 168)
SPIN(spin_512_168, 512, 168)
SPIN(spin_768_128, 768, 128)
SPIN(spin_1024_128, 1024, 128)
SPIN(spin_256_128, 256, 128)

typedef void (*kern_t)(unsigned long long*, int);
void run(kern_t k, int THREADS, int VGPRS, int lds_kb, int grid) {
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 4 * grid);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    const int ticks = 100 * 100;   // wall_clock64 ticks at 100 MHz: 100 us
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ev_ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {   // (the event pair of the last launch: what a profiler calls the kernel's duration)
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds_kb * 1024, 0, d, ticks);
        hipEventRecord(e1, 0);
    }
    hipDeviceSynchronize();
    hipEventElapsedTime(&ev_ms, e0, e1);
    std::vector<unsigned long long> h(4 * grid);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    std::set<unsigned long long> cus;
    int late = 0;
    for (int i = 0; i < grid; ++i) tmin = std::min(tmin, h[4 * i]);
    for (int i = 0; i < grid; ++i) {
        tmax = std::max(tmax, h[4 * i + 1]);
        if (h[4 * i] - tmin > 5000) ++late;                    // started more than 50 us after the first: a second round
        const unsigned hw = (unsigned)h[4 * i + 2];
        cus.insert(((h[4 * i + 2] >> 32) << 16) | (((hw >> 13) & 0x7) << 8) | ((hw >> 8) & 0xf));   // (xcc, se, cu)
    }
    printf("threads %4d vgprs %3d lds %3d KiB grid %3d: first start -> last end %6.1f us, event pair %6.1f us, %3d workgroups started late, %3zu distinct (xcc, se, cu)\n",
           THREADS, VGPRS, lds_kb, grid, (tmax - tmin) / 100.0, ev_ms * 1e3, late, cus.size());
    hipFree(d);
}

int main() {
    for (int grid : {200, 208, 210, 211, 212, 216, 232, 256, 264}) run(spin_768_168, 768, 168, 96, grid);
    for (int grid : {208, 216, 256}) run(spin_512_168, 512, 168, 96, grid);
    for (int grid : {208, 216, 256}) run(spin_768_168, 768, 168, 64, grid);
    for (int grid : {208, 216, 256}) run(spin_768_168, 768, 168, 80, grid);
    for (int grid : {208, 216, 256, 264}) run(spin_1024_128, 1024, 128, 96, grid);
    for (int grid : {256, 264, 512, 520}) run(spin_256_128, 256, 128, 48, grid);
    return 0;
}
