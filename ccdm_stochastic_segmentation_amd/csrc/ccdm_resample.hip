// The two resamplers of the `resblock_updown=True` topology (/root/reference/ddpm/models/unet_openai/unet.py:202-219, :242-250): a
// ResBlock(down=True) runs  h = in_conv(AvgPool2d(2)(SiLU(GroupNorm(x)))),  x = AvgPool2d(2)(x);  a ResBlock(up=True) the same with a
// nearest x2 upsample (its in_conv upsamples on load — ccdm_conv_args.up — so only the raw branch comes through here).  The pool sits
// BETWEEN the activation and the conv, so it cannot ride in the conv's halo staging the way GroupNorm + SiLU do; it is one elementwise
// pass that reads x once and writes both quarter-size tensors.  HBM-bound: 4 C bytes in, 2 C bytes out per input pixel.
//   AvgPool2d(2): ((x00 + x01) + x10) + x11, then * 0.25 — torch's CPU order (rows outer, columns inner), so the raw branch is bit-exact.
#include "ccdm_common.h"
#include "ccdm_conv_common.h"

namespace ccdm {

struct ResampleK {
    ccdm_conv_args g;          // in0 / C0 / stats0 / slices0 / gamma / beta / eps / act / N / Hin / Win / Hout / Wout; the rest zero
    float* out_act;
    float* out_raw;
    int mode;
};

__device__ __forceinline__ float4 resample_act(const float4 v, const float2* ab, const int c, const bool has_gn, const bool silu) {
    float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float y = x[j];
        if (has_gn) y = fmaf(y, ab[c + j].x, ab[c + j].y);
        if (silu) y = y / (1.0f + expf(-y));
        x[j] = y;
    }
    return make_float4(x[0], x[1], x[2], x[3]);
}

__device__ __forceinline__ float4 pool4(const float4 a, const float4 b, const float4 c, const float4 d) {
    return make_float4((((a.x + b.x) + c.x) + d.x) * 0.25f, (((a.y + b.y) + c.y) + d.y) * 0.25f,
                       (((a.z + b.z) + c.z) + d.z) * 0.25f, (((a.w + b.w) + c.w) + d.w) * 0.25f);
}

__global__ __launch_bounds__(256) void k_resample(const ResampleK k) {
    extern __shared__ __attribute__((aligned(16))) char smem_rs[];
    float2* ab = reinterpret_cast<float2*>(smem_rs);                 // [C] GroupNorm (scale, shift) of sample n
    const ccdm_conv_args& a = k.g;
    const int n = blockIdx.y, C = a.C0, Q = C >> 2;
    const bool has_gn = a.stats0 != nullptr && k.out_act != nullptr, silu = a.act == CCDM_ACT_SILU;
    if (has_gn) {
        compute_gn_affine(a, n, 0, ab);
        __syncthreads();
    }
    const int Ho = a.Hout, Wo = a.Wout;
    const float* in = a.in0 + (size_t)n * a.Hin * a.Win * C;
    float* oa = k.out_act ? k.out_act + (size_t)n * Ho * Wo * C : nullptr;
    float* orw = k.out_raw ? k.out_raw + (size_t)n * Ho * Wo * C : nullptr;
    const long long items = (long long)Ho * Wo * Q;
    for (long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x; item < items; item += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(item % Q), p = (int)(item / Q), ox = p % Wo, oy = p / Wo, c = 4 * q;
        const size_t o = (size_t)p * C + c;
        if (k.mode == CCDM_RESAMPLE_AVGPOOL2) {
            const float* r0 = in + ((size_t)(2 * oy) * a.Win + 2 * ox) * C + c;
            const float* r1 = r0 + (size_t)a.Win * C;
            const float4 v00 = *reinterpret_cast<const float4*>(r0), v01 = *reinterpret_cast<const float4*>(r0 + C);
            const float4 v10 = *reinterpret_cast<const float4*>(r1), v11 = *reinterpret_cast<const float4*>(r1 + C);
            if (orw) *reinterpret_cast<float4*>(orw + o) = pool4(v00, v01, v10, v11);
            if (oa)
                *reinterpret_cast<float4*>(oa + o) = pool4(resample_act(v00, ab, c, has_gn, silu), resample_act(v01, ab, c, has_gn, silu),
                                                           resample_act(v10, ab, c, has_gn, silu), resample_act(v11, ab, c, has_gn, silu));
        } else {
            const float4 v = *reinterpret_cast<const float4*>(in + ((size_t)(oy >> 1) * a.Win + (ox >> 1)) * C + c);
            if (orw) *reinterpret_cast<float4*>(orw + o) = v;
            if (oa) *reinterpret_cast<float4*>(oa + o) = resample_act(v, ab, c, has_gn, silu);
        }
    }
}

int launch_resample(const ccdm_resample_args& r, hipStream_t s) {
    CCDM_REQUIRE(r.in && (r.out_act || r.out_raw), "resample: null pointer");
    CCDM_REQUIRE(r.mode == CCDM_RESAMPLE_AVGPOOL2 || r.mode == CCDM_RESAMPLE_NEAREST_UP2, "resample: mode = %d", r.mode);
    CCDM_REQUIRE(r.C > 0 && r.C % 4 == 0 && r.N > 0 && r.Hin > 0 && r.Win > 0, "resample: N=%d C=%d %dx%d", r.N, r.C, r.Hin, r.Win);
    CCDM_REQUIRE(r.act == CCDM_ACT_NONE || r.act == CCDM_ACT_SILU, "resample: act = %d", r.act);
    ResampleK k{};
    k.g.in0 = r.in; k.g.C0 = r.C;
    k.g.N = r.N; k.g.Hin = r.Hin; k.g.Win = r.Win;
    k.g.Hout = r.mode == CCDM_RESAMPLE_AVGPOOL2 ? r.Hin / 2 : 2 * r.Hin;
    k.g.Wout = r.mode == CCDM_RESAMPLE_AVGPOOL2 ? r.Win / 2 : 2 * r.Win;
    CCDM_REQUIRE(k.g.Hout > 0 && k.g.Wout > 0, "resample: AvgPool2d(2) of a %dx%d image is empty", r.Hin, r.Win);
    k.g.act = r.act; k.g.eps = r.eps; k.g.emb_off = -1;
    if (r.stats) {
        CCDM_REQUIRE(r.out_act, "resample: GroupNorm statistics without an out_act destination");
        CCDM_REQUIRE(r.C % 32 == 0 && r.C <= CCDM_MAX_CHANNELS, "resample: GroupNorm(32, %d)", r.C);
        CCDM_REQUIRE(r.gamma && r.beta && r.slices >= 1 && r.slices <= CCDM_STATS_MAX_SLICES, "resample: GroupNorm needs gamma/beta and 1..%d slices (got %d)",
                     CCDM_STATS_MAX_SLICES, r.slices);
        k.g.stats0 = r.stats; k.g.slices0 = r.slices; k.g.gamma = r.gamma; k.g.beta = r.beta;
    }
    k.out_act = r.out_act; k.out_raw = r.out_raw; k.mode = r.mode;
    const long long items = (long long)k.g.Hout * k.g.Wout * (r.C / 4);
    // >> 256 workgroups over the batch where the image allows it; a block covers at least 4 items per thread
    const long long want = (items + 1023) / 1024;
    const int bx = (int)(want < 1 ? 1 : (want > 64 ? 64 : want));
    hipLaunchKernelGGL(k_resample, dim3(bx, r.N), dim3(256), r.stats ? (size_t)r.C * 8 : 0, s, k);
    CCDM_CHECK_LAUNCH("resample");
    return 0;
}

}  // namespace ccdm

extern "C" int ccdm_resample(const ccdm_resample_args* a, void* stream) {
    if (!a) return ccdm::fail("ccdm_resample: null args");
    return ccdm::launch_resample(*a, (hipStream_t)stream);
}
