// Do MFMAs of one wave and VALU work of another wave on the SAME SIMD overlap on gfx950?
// 512-thread blocks, one per CU (100 KB LDS): waves 0-3 sit on SIMD 0-3 and run an MFMA loop (MF),
// waves 4-7 sit on the same SIMDs and run a VALU loop (VA = 1: fma, 2: exp2+add+rcp, 3: ds_read_b128 stream).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MF, int VA>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ char smem[];
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (MF) {
            f32x16 a0 = {}, a1 = {};
            f16x8 x, y;
            for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f); }
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (MF == 2) {      // accumulators pinned to the AccVGPR file
                        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(a0) : "v"(x), "v"(y));
                        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(a1) : "v"(y), "v"(x));
                    } else {
                        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
                    }
                }
            }
            r = a0[0] + a1[3];
        }
    } else if (VA == 3) {
        const f32x4* p = reinterpret_cast<const f32x4*>(smem) + (threadIdx.x & 255);
        f32x4 s = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) s += p[u * 256];
        }
        r = s[0] + s[1] + s[2] + s[3];
    } else if (VA) {
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 0.01f + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (VA == 2) v[j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[j]));
                    else v[j] = fmaf(v[j], 1.0001f, 0.5f);
                }
        }
        for (int j = 0; j < 16; ++j) r += v[j];
    }
    if (r == 123.456f) out[threadIdx.x] = r + smem[0];
}
template <int MF, int VA> void run(float* d, int iters, const char* name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MF, VA>), dim3(256), dim3(512), 100 * 1024, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-34s %8.1f us\n", name, ms * 1e3);
}
int main() {
    float* d; (void)hipMalloc(&d, 4096);
    const int iters = 2000;
    printf("per SIMD: MFMA wave = %d MFMA 32x32x16 (32 cycles each: %.0f us at 2.4 GHz); VALU wave = %d fma | %d (exp2,add,rcp) | %d ds_read_b128\n",
           iters * 16, iters * 16 * 32 / 2400.0, iters * 128, iters * 128, iters * 16);
    run<1, 0>(d, iters, "mfma only");
    run<0, 1>(d, iters, "valu(fma) only");
    run<1, 1>(d, iters, "mfma + valu(fma)");
    run<0, 2>(d, iters, "valu(exp2+add+rcp) only");
    run<1, 2>(d, iters, "mfma + valu(exp2+add+rcp)");
    run<2, 0>(d, iters, "mfma(AGPR acc) only");
    run<2, 1>(d, iters, "mfma(AGPR acc) + valu(fma)");
    run<2, 3>(d, iters, "mfma(AGPR acc) + lds read");
    run<0, 3>(d, iters, "lds read only");
    run<1, 3>(d, iters, "mfma + lds read");
    return 0;
}
