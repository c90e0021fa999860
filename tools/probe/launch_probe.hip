// What a chain of dependent tiny kernels costs per kernel on one stream (the floor under the ViT's 63 launches), and the same chain as a graph.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_probe tools/probe/launch_probe.hip && /tmp/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0f; }
__global__ void wide(float* x, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) x[i] += 1.0f; }
int main() {
    float* x;
    hipMalloc(&x, 64 << 20);
    hipMemset(x, 0, 64 << 20);
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) {
        const int grid = mode == 0 ? 1 : mode == 1 ? 256 : mode == 2 ? 2048 : 16384;   // 1 block; 1 per CU; 8 per CU; 16 MB touched
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, st);
            for (int i = 0; i < 630; ++i) {
                if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, x);
                else hipLaunchKernelGGL(wide, dim3(grid), dim3(256), 0, st, x, grid * 256);
            }
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("chain of 630 kernels, grid %5d x 256: %.2f us per kernel\n", grid, ms * 1e3 / 630);
        }
    }
    return 0;
}
