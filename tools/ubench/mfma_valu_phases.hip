// Round 6: WHY do the matrix instructions of one wave and the vector instructions of another wave on the same SIMD add up
// (mfma_valu_overlap.hip) when the same mix inside one wave overlaps (mfma_valu_interleave.hip)?  Hypothesis: a wave whose next
// instruction is an MFMA keeps requesting the SIMD's VALU issue port while the matrix pipe is busy, and the arbiter (priority, then
// age) serves it first — the other waves' plain VALU instructions never see the 28 idle issue cycles behind each MFMA.
//
// Part 1 (pure streams, as mfma_valu_overlap): 512-thread blocks, one per CU; waves 0-3 = MFMA stream, waves 4-7 = fma stream (or the
// other way round, ORDER = 1).  Knobs: NOP = number of `s_nop 7` between two MFMAs of the MFMA wave; PV / PM = s_setprio of the VALU /
// MFMA wave.
// Part 2 (the conv kernel's shape): 256-thread blocks, 3 per CU (48 KB LDS each), every wave alternates a VALU phase (VF fma) and a
// matrix phase (54 MFMAs with `NOP` nops in between), blocks start with different phase offsets; PV = priority during the VALU phase.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NOP> __device__ __forceinline__ void nops() {
    if constexpr (NOP >= 1) asm volatile("s_nop 7");
    if constexpr (NOP >= 2) asm volatile("s_nop 7");
    if constexpr (NOP >= 3) asm volatile("s_nop 7");
}
template <int P> __device__ __forceinline__ void setprio() {
    if constexpr (P == 1) asm volatile("s_setprio 1");
    if constexpr (P == 2) asm volatile("s_setprio 2");
    if constexpr (P == 3) asm volatile("s_setprio 3");
}

template <int MF, int VA, int NOP, int PV, int PM, int ORDER>
__global__ __launch_bounds__(512) void k1(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const bool is_m = ORDER ? wave >= 4 : wave < 4;
    float r = 0.f;
    if (is_m) {
        if (MF) {
            setprio<PM>();
            f32x16 a0 = {}, a1 = {};
            f16x8 x, y;
            for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f); }
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
                    nops<NOP>();
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a1) : "v"(y), "v"(x));
                    nops<NOP>();
                }
            }
            r = a0[0] + a1[3];
        }
    } else if (VA) {
        setprio<PV>();
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 0.01f + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(1.0001f), "v"(0.5f));
        }
        for (int j = 0; j < 16; ++j) r += v[j];
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}

// Part 2: phases.  VF = fma per VALU phase (in units of 16), 54 MFMAs per matrix phase.
template <int MF, int VF, int NOP, int PV>
__global__ __launch_bounds__(256, 3) void k2(float* out, int iters) {
    extern __shared__ char smem[];
    f32x16 a0 = {}, a1 = {};
    f16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(threadIdx.x * 0.001f + j); y[j] = (_Float16)(j * 0.5f); }
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 0.01f + j;
    // de-synchronise the three blocks of a CU: block b starts with (b / 256) thirds of a VALU phase
    const int pre = (blockIdx.x / 256) * (VF / 3);
    for (int u = 0; u < pre; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(1.0001f), "v"(0.5f));
    for (int i = 0; i < iters; ++i) {
        if (VF) {
            setprio<PV>();
#pragma unroll 4
            for (int u = 0; u < VF; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(1.0001f), "v"(0.5f));
            if (PV) asm volatile("s_setprio 0");
        }
        if (MF) {
#pragma unroll
            for (int u = 0; u < 27; ++u) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
                nops<NOP>();
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a1) : "v"(y), "v"(x));
                nops<NOP>();
            }
        }
    }
    float r = a0[0] + a1[3];
    for (int j = 0; j < 16; ++j) r += v[j];
    if (r == 123.456f) out[threadIdx.x] = r + smem[0];
}

static float* d;
template <int MF, int VA, int NOP, int PV, int PM, int ORDER> void run1(const char* name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k1<MF, VA, NOP, PV, PM, ORDER>), dim3(256), dim3(512), 100 * 1024, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("P1 %-58s %8.1f us\n", name, ms * 1e3);
}
template <int MF, int VF, int NOP, int PV> void run2(const char* name) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    const int iters = 300;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k2<MF, VF, NOP, PV>), hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k2<MF, VF, NOP, PV>), dim3(768), dim3(256), 48 * 1024, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("P2 %-58s %8.1f us\n", name, ms * 1e3);
}
int main() {
    (void)hipMalloc(&d, 1 << 16);
    run1<1, 0, 0, 0, 0, 0>("mfma only");
    run1<1, 0, 1, 0, 0, 0>("mfma only, 1 x s_nop 7 between");
    run1<1, 0, 2, 0, 0, 0>("mfma only, 2 x s_nop 7 between");
    run1<1, 0, 3, 0, 0, 0>("mfma only, 3 x s_nop 7 between");
    run1<0, 1, 0, 0, 0, 0>("fma only");
    run1<1, 1, 0, 0, 0, 0>("mfma + fma");
    run1<1, 1, 0, 0, 0, 1>("mfma + fma, VALU waves older");
    run1<1, 1, 0, 3, 0, 0>("mfma + fma, VALU prio 3");
    run1<1, 1, 0, 0, 3, 0>("mfma + fma, MFMA prio 3");
    run1<1, 1, 0, 3, 0, 1>("mfma + fma, VALU prio 3, VALU waves older");
    run1<1, 1, 1, 0, 0, 0>("mfma(1 nop) + fma");
    run1<1, 1, 2, 0, 0, 0>("mfma(2 nop) + fma");
    run1<1, 1, 3, 0, 0, 0>("mfma(3 nop) + fma");
    run1<1, 1, 2, 3, 0, 0>("mfma(2 nop) + fma, VALU prio 3");
    run1<1, 1, 3, 3, 0, 0>("mfma(3 nop) + fma, VALU prio 3");
    // phases: 54 MFMAs (1728 pipe cycles) + VF*16 fma per iteration and wave, 3 waves per SIMD
    run2<1, 0, 0, 0>("3 blk/CU: matrix phases only");
    run2<0, 24, 0, 0>("3 blk/CU: VALU phases only (384 fma)");
    run2<1, 24, 0, 0>("3 blk/CU: both");
    run2<1, 24, 0, 3>("3 blk/CU: both, VALU phase prio 3");
    run2<1, 24, 0, 1>("3 blk/CU: both, VALU phase prio 1");
    run2<1, 24, 1, 0>("3 blk/CU: both, 1 nop between MFMAs");
    run2<1, 24, 2, 0>("3 blk/CU: both, 2 nop between MFMAs");
    run2<1, 24, 1, 3>("3 blk/CU: both, 1 nop, VALU prio 3");
    run2<1, 24, 2, 3>("3 blk/CU: both, 2 nop, VALU prio 3");
    run2<1, 48, 0, 0>("3 blk/CU: both, 768 fma per phase");
    run2<1, 48, 0, 3>("3 blk/CU: both, 768 fma per phase, VALU prio 3");
    run2<0, 48, 0, 0>("3 blk/CU: VALU only, 768 fma per phase");
    return 0;
}
