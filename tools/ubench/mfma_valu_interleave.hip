// Does plain VALU work placed BETWEEN the MFMAs of the SAME wave hide in the matrix pipe's 32-cycle issue interval on gfx950?
// (tools/ubench/mfma_valu_overlap.hip showed that MFMAs of one wave and VALU work of ANOTHER wave on the same SIMD add up.)
// W waves per SIMD (blocks of 256*W threads, one block per CU), every wave runs 16 x { v_mfma_f32_32x32x16_f16 ; K x v_fma_f32 }
// per trip, two accumulators alternating, order pinned with inline asm.  TR = 1: fillers are v_exp_f32 (quarter rate).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int K, int TR>
__global__ void k(float* out, int iters) {
    f32x16 a0 = {}, a1 = {};
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (TR) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 7]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(1.0001f), "v"(0.5f));
            }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a1) : "v"(y), "v"(x));
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (TR) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(j + 4) & 7]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(j + 4) & 7]) : "v"(1.0001f), "v"(0.5f));
            }
        }
    }
    float r = a0[0] + a1[3];
    for (int j = 0; j < 8; ++j) r += v[j];
    if (r == 123.456f) out[threadIdx.x] = r;
}
template <int K, int TR> void run(float* d, int iters, int W) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<K, TR>), dim3(256), dim3(256 * W), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double mf = (double)iters * 16 * W;          // MFMAs per SIMD
    printf("W=%d K=%d %s: %8.1f us  = %5.1f cycles per MFMA slot at 2.0 GHz (32 = the matrix pipe alone)\n", W, K, TR ? "exp" : "fma", ms * 1e3, ms * 1e-3 * 2.0e9 / mf);
}
int main() {
    float* d; (void)hipMalloc(&d, 1 << 16);
    const int iters = 1000;
    for (int W = 1; W <= 3; W += 2) {
        run<0, 0>(d, iters, W); run<2, 0>(d, iters, W); run<4, 0>(d, iters, W); run<5, 0>(d, iters, W); run<6, 0>(d, iters, W); run<8, 0>(d, iters, W);
        run<12, 0>(d, iters, W); run<2, 1>(d, iters, W); run<4, 1>(d, iters, W);
    }
    return 0;
}
