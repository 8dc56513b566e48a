// Sustained fp32-MFMA rate probe: what the chip actually delivers under a long dense
// v_mfma_f32_32x32x2_f32 stream (clock under load), to calibrate roofline fractions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * NACC * 4096.0;
    printf("NACC=%d blocks=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, iters, ms, flop / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    run<4>(256, 20000, out);     // 1 wave / SIMD, ~20 ms
    run<4>(512, 20000, out);     // 2 waves / SIMD
    run<4>(1024, 20000, out);    // 4 waves / SIMD
    run<8>(256, 20000, out);
    run<4>(256, 2000, out);      // short kernel (~2 ms)
    run<4>(256, 200, out);       // very short
    return 0;
}
