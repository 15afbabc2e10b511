// LIDC evaluation metrics, device part: per-class intersection / union pixel counts between every
// pair (sample i, reference j) of class-index maps of one image.  The reference builds a [B,S,S',HW,K] boolean
// broadcast on the host (evaluation/evaluate_lidc_uncertainty.py:27-39, `batched_distance`); here one block per
// (image, i, j) streams the two byte maps once and counts in registers.  Counts are exact integers, so the IoU,
// GED and Hungarian-matched IoU the host derives from them are bit-identical to the reference's numpy result.
#include "ccdm_common.h"

namespace ccdm {

template <int KP>
__global__ __launch_bounds__(256) void k_pair_counts(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                     int S, int L, int HW, int K, int32_t* __restrict__ out) {
    __shared__ int red[4][KP][2];
    const int j = blockIdx.x % L, i = (blockIdx.x / L) % S, img = blockIdx.x / (L * S);
    const uint8_t* pa = a + ((size_t)img * S + i) * HW;
    const uint8_t* pb = b + ((size_t)img * L + j) * HW;
    int inter[KP], uni[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { inter[k] = 0; uni[k] = 0; }
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        const int ca = pa[p], cb = pb[p];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const bool xa = ca == k, xb = cb == k;
            inter[k] += (xa & xb) ? 1 : 0;
            uni[k] += (xa | xb) ? 1 : 0;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        int x = inter[k], u = uni[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { x += __shfl_xor(x, off); u += __shfl_xor(u, off); }
        if (lane == 0) { red[wave][k][0] = x; red[wave][k][1] = u; }
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        int x = 0, u = 0;
        for (int w = 0; w < 4; ++w) { x += red[w][k][0]; u += red[w][k][1]; }
        int32_t* o = out + ((size_t)blockIdx.x * K + k) * 2;
        o[0] = x; o[1] = u;
    }
}

}  // namespace ccdm

extern "C" int ccdm_pairwise_class_counts(const uint8_t* a, const uint8_t* b, int B, int S, int L, int HW, int K,
                                          int32_t* out, void* stream) {
    using namespace ccdm;
    CCDM_REQUIRE(a && b && out, "pairwise_class_counts: null pointer");
    CCDM_REQUIRE(K >= 1 && K <= 32, "pairwise_class_counts: K=%d outside [1,32]", K);
    const int blocks = B * S * L;
    if (blocks <= 0 || HW <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (K <= 2) hipLaunchKernelGGL(k_pair_counts<2>, dim3(blocks), dim3(256), 0, s, a, b, S, L, HW, K, out);
    else if (K <= 8) hipLaunchKernelGGL(k_pair_counts<8>, dim3(blocks), dim3(256), 0, s, a, b, S, L, HW, K, out);
    else hipLaunchKernelGGL(k_pair_counts<32>, dim3(blocks), dim3(256), 0, s, a, b, S, L, HW, K, out);
    CCDM_CHECK_LAUNCH("pairwise_class_counts");
    return 0;
}
