// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) operands on gfx950: operand layout hypothesis + issue rate.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mx6_probe tools/probe/mx6_probe.hip && gpurun_out/mx6_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

static float f6_value(unsigned c) {   // e2m3: sign 1, exponent 2 (bias 1), mantissa 3
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -v : v;
}

// hypothesis: lane l holds row (l & 31), k = 32 (l >> 5) + f, f = 0..31, field f = bits [6f, 6f + 6) of the lane's 192-bit operand
__global__ void probe(const unsigned char* A6, const unsigned char* B6, const unsigned char* sa, const unsigned char* sb, float* D) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    unsigned wa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int f = 0; f < 32; ++f) {
        const unsigned ca = A6[r * 64 + 32 * h + f], cb = B6[r * 64 + 32 * h + f];   // A[row r][k], B^T[col r][k]
        const int bit = 6 * f, w = bit >> 5, o = bit & 31;
        wa[w] |= ca << o;
        if (o > 26) wa[w + 1] |= ca >> (32 - o);
        wb[w] |= cb << o;
        if (o > 26) wb[w + 1] |= cb >> (32 - o);
    }
    intx8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)wa[i]; b[i] = (int)wb[i]; }
    floatx16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    const int scale_a = sa[r * 2 + h], scale_b = sb[r * 2 + h];   // E8M0 of (row, 32-element block)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, scale_a, 0, scale_b);
    for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = c[e];   // D[row of A][col = row of B^T]
}

__global__ void rate(float* out, int iters) {
    intx8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7 + i; b[i] = threadIdx.x * 13 + i; }
    floatx16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = c2[i] = c3[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 2, 2, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 2, 2, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 2, 2, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 2, 2, 0, 127, 0, 127);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

// round 5: the K = 128 shape of the same instruction family, v_mfma_scale_f32_16x16x128_f8f6f4 -- 16 x 16 x 128
// multiply-adds per instruction (half of 32 x 32 x 64's), four result registers per lane
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void rate16(float* out, int iters) {
    intx8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7 + i; b[i] = threadIdx.x * 13 + i; }
    floatx4 c[8];
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 4; ++i) c[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[j], 2, 2, 0, 127, 0, 127);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += c[j][j & 3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<unsigned char> A(32 * 64), B(32 * 64), sa(64), sb(64);
    srand(1);
    for (auto& x : A) x = rand() & 63;
    for (auto& x : B) x = rand() & 63;
    for (auto& x : sa) x = 127 + (rand() % 5) - 2;
    for (auto& x : sb) x = 127 + (rand() % 5) - 2;
    unsigned char *dA, *dB, *dsa, *dsb;
    float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice);
    hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dsa, dsb, dD);
    std::vector<float> D(32 * 32);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int k = 0; k < 64; ++k)
                ref += (double)f6_value(A[i * 64 + k]) * ldexp(1.0, sa[i * 2 + k / 32] - 127) * (double)f6_value(B[j * 64 + k]) * ldexp(1.0, sb[j * 2 + k / 32] - 127);
            worst = fmax(worst, fabs(ref - D[i * 32 + j]));
        }
    printf("layout hypothesis: max |D - ref| = %g (D[0][0] = %g, D[3][5] = %g)\n", worst, D[0], D[3 * 32 + 5]);
    float* dout;
    hipMalloc(&dout, (size_t)1024 * 1024 * 4);   // (the largest launch below: 1024 threads x 1024 workgroups)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    rate<<<1024, 256>>>(dout, 100);
    hipEventRecord(e0);
    rate<<<1024, 256>>>(dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 1024.0 * 4 * iters * 4 * 2.0 * 32 * 32 * 64;
    printf("rate: %.1f TFLOP/s (fp6 x fp6, 4 accumulators per wave, 4 waves per workgroup, 1024 workgroups)\n", flops / (ms * 1e-3) / 1e12);
    // the same instruction at other occupancies (waves per SIMD: 1 / 2 / 4), then the 16 x 16 x 128 shape
    for (int threads : {256, 512, 1024}) {
        for (int wgs : {256, 1024}) {
            rate<<<wgs, threads>>>(dout, 100);
            hipEventRecord(e0);
            rate<<<wgs, threads>>>(dout, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("32x32x64  %4d threads x %4d workgroups: %.1f TFLOP/s\n", threads, wgs,
                   (double)wgs * (threads / 64) * iters * 4 * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12);
            rate16<<<wgs, threads>>>(dout, 100);
            hipEventRecord(e0);
            rate16<<<wgs, threads>>>(dout, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("16x16x128 %4d threads x %4d workgroups: %.1f TFLOP/s\n", threads, wgs,
                   (double)wgs * (threads / 64) * iters * 8 * 2.0 * 16 * 16 * 128 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
