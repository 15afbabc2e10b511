// How fast does ONE dependent accumulation chain of v_mfma_f32_32x32x16_f16 run on gfx950 (every MFMA reads the accumulator the previous
// one wrote), against CH independent chains issued round robin?  W waves per SIMD, one block per CU.  A dependent MFMA that is a wave's
// only ready instruction leaves the matrix pipe idle until its predecessor retires; another wave (or another chain) fills the gap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int CH, int RND = 0>
__global__ void k(float* out, int iters) {
    f32x16 a[CH];
    for (int c = 0; c < CH; ++c) a[c] = f32x16{};
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f); }
    if (RND) {        // operands with random mantissas and signs (data-dependent power: do the clocks hold?)
        unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u; x[j] = (_Float16)(((int)(h >> 8) & 0xFFFF) * (1.0f / 32768.0f) - 1.0f);
            h = h * 1664525u + 1013904223u; y[j] = (_Float16)(((int)(h >> 8) & 0xFFFF) * (1.0f / 32768.0f) - 1.0f);
        }
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 12; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a[u % CH]) : "v"(x), "v"(y));
    }
    float r = 0;
    for (int c = 0; c < CH; ++c) r += a[c][c];
    if (r == 123.456f) out[threadIdx.x] = r;
}
template <int CH, int RND = 0> void run(float* d, int iters, int W) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<CH, RND>), dim3(256), dim3(256 * W), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double mf = (double)iters * 12 * W;          // MFMAs per SIMD
    printf("W=%d chains=%d %s: %8.1f us = %5.1f cycles per MFMA at 2.0 GHz (32 = the matrix pipe's issue interval)\n", W, CH, RND ? "random operands" : "smooth operands", ms * 1e3, ms * 1e-3 * 2.0e9 / mf);
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 20);
    for (int W = 1; W <= 3; ++W) { run<1>(d, 20000 / W, W); run<2>(d, 20000 / W, W); run<3>(d, 20000 / W, W); run<4>(d, 20000 / W, W); }
    for (int W = 1; W <= 3; ++W) { run<1, 1>(d, 20000 / W, W); run<2, 1>(d, 20000 / W, W); run<4, 1>(d, 20000 / W, W); }
    run<2, 1>(d, 400000, 2);       // long enough (~70 ms) for the power management to react
    run<2, 0>(d, 400000, 2);
    return 0;
}
