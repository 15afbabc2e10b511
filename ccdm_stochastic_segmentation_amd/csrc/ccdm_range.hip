// F16X3 range diagnostics: the largest |a| a conv stages, i.e. what the fp16 hi/lo split of ccdm_conv.hip / ccdm_conv_ks.hip /
// ccdm_conv1x1.hip sees of its inputs BEFORE the 2^4 pre-scale — GroupNorm (+ SiLU) applied to the main input where the conv
// normalises on load, raw values otherwise, and the raw input of a fused 1x1 skip segment.  The split is exact to 2^-22 for
// |a| < 4094 (include/ccdm_hip.h); beyond it the hi half overflows.  Used by tools/range_report.py (per-layer headroom of a
// state_dict) and by the host's per-layer fp32 fallback; not on the sampling path.
//   max is order-independent: one atomicMax on the value's bit pattern (non-negative floats order like unsigned integers).
#include "ccdm_common.h"
#include "ccdm_conv_common.h"

namespace ccdm {

__global__ __launch_bounds__(256) void k_conv_input_absmax(const ccdm_conv_args a, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem_r[];
    float2* ab = reinterpret_cast<float2*>(smem_r);                  // [C] GroupNorm (scale, shift) of sample n
    const int n = blockIdx.y, C = a.C0 + a.C1;
    const bool has_gn = a.stats0 != nullptr;
    const int step = a.step_ptr ? *a.step_ptr : 0;
    const int emb_row = (a.emb_row_of_sample ? a.emb_row_of_sample[n] : 0) + step;
    if (has_gn) {
        compute_gn_affine(a, n, emb_row, ab);
        __syncthreads();
    }
    const int HW = a.Hin * a.Win, Q = C >> 2;
    float m = 0.f;
    bool bad = false;
    for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < (long long)HW * Q; item += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(item / Q), c = 4 * (int)(item % Q);
        const bool second = c >= a.C0;
        const float* src = second ? a.in1 + ((size_t)n * HW + p) * a.C1 + (c - a.C0) : a.in0 + ((size_t)n * HW + p) * a.C0 + c;
        const float4 v = *reinterpret_cast<const float4*>(src);
        float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float y = x[j];
            if (has_gn) y = fmaf(y, ab[c + j].x, ab[c + j].y);
            if (a.act == CCDM_ACT_SILU) y = y / (1.0f + expf(-y));
            bad |= !(fabsf(y) <= 3.0e38f);
            m = fmaxf(m, fabsf(y));
        }
    }
    if (a.skip0) {
        const int HWo = a.Hout * a.Wout, SC = a.SC0 + a.SC1, Qs = SC >> 2;
        for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < (long long)HWo * Qs; item += (long long)gridDim.x * blockDim.x) {
            const int p = (int)(item / Qs), c = 4 * (int)(item % Qs);
            const bool second = c >= a.SC0;
            const float* src = second ? a.skip1 + ((size_t)n * HWo + p) * a.SC1 + (c - a.SC0) : a.skip0 + ((size_t)n * HWo + p) * a.SC0 + c;
            const float4 v = *reinterpret_cast<const float4*>(src);
            bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    }
    if (bad) m = __builtin_inff();                                   // a NaN / Inf input reads as "beyond any range"
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
}

int launch_conv_input_absmax(const ccdm_conv_args& a, float* out, hipStream_t s) {
    CCDM_REQUIRE(a.in0 && out, "conv_input_absmax: null pointer");
    const int C = a.C0 + a.C1;
    CCDM_REQUIRE(C > 0 && a.C0 % 4 == 0 && a.C1 % 4 == 0 && C <= CCDM_MAX_CHANNELS, "conv_input_absmax: C0=%d C1=%d", a.C0, a.C1);
    const long long items = (long long)a.Hin * a.Win * (C / 4);
    const int bx = (int)(items / 256 < 1 ? 1 : (items / 256 > 256 ? 256 : items / 256));
    hipLaunchKernelGGL(k_conv_input_absmax, dim3(bx, a.N), dim3(256), (size_t)C * 8, s, a, out);
    CCDM_CHECK_LAUNCH("conv_input_absmax");
    return 0;
}

}  // namespace ccdm

extern "C" int ccdm_conv_input_absmax(const ccdm_conv_args* a, float* out, void* stream) {
    if (!a) return ccdm::fail("ccdm_conv_input_absmax: null args");
    return ccdm::launch_conv_input_absmax(*a, out, (hipStream_t)stream);
}
