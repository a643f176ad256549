// Micro-benchmark: sustained issue rate of the fp32 MFMA shapes on gfx950, independent accumulators, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static double run(F launch, double flops) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return flops * 5 / (ms * 1e-3) * 1e-12;
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD (block = 4 * wps waves, one block per CU)
        const int threads = 256 * wps, blocks = 256;
        const double waves = (double)blocks * threads / 64;
        printf("waves/SIMD %d\n", wps);
        printf("  16x16x4  x16 acc : %6.1f TFLOP/s\n", run([&] { k16<16><<<blocks, threads>>>(out, iters, 1.f, 2.f); }, waves * iters * 16 * 2048.0));
        printf("  16x16x4  x4 acc  : %6.1f TFLOP/s\n", run([&] { k16<4><<<blocks, threads>>>(out, iters, 1.f, 2.f); }, waves * iters * 4 * 2048.0));
        printf("  16x16x4  x2 acc  : %6.1f TFLOP/s\n", run([&] { k16<2><<<blocks, threads>>>(out, iters, 1.f, 2.f); }, waves * iters * 2 * 2048.0));
        printf("  32x32x2  x4 acc  : %6.1f TFLOP/s\n", run([&] { k32<4><<<blocks, threads>>>(out, iters, 1.f, 2.f); }, waves * iters * 4 * 4096.0));
        printf("  32x32x2  x2 acc  : %6.1f TFLOP/s\n", run([&] { k32<2><<<blocks, threads>>>(out, iters, 1.f, 2.f); }, waves * iters * 2 * 4096.0));
        printf("  32x32x2  x1 acc  : %6.1f TFLOP/s\n", run([&] { k32<1><<<blocks, threads>>>(out, iters, 1.f, 2.f); }, waves * iters * 1 * 4096.0));
    }
    return 0;
}
