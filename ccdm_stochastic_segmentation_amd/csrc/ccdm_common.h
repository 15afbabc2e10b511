// Shared device/host helpers for libccdm_hip.so (gfx950 only).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "ccdm_hip.h"

namespace ccdm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

void set_error(const std::string& s);
int fail(const char* fmt, ...);

#define CCDM_CHECK_LAUNCH(what)                                               \
    do {                                                                      \
        hipError_t _e = hipGetLastError();                                    \
        if (_e != hipSuccess) return ::ccdm::fail("%s: %s", what, hipGetErrorString(_e)); \
    } while (0)

#define CCDM_REQUIRE(cond, ...)                     \
    do {                                            \
        if (!(cond)) return ::ccdm::fail(__VA_ARGS__); \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Experiment switches (same-box A/B hooks of tools/): environment variables are read ONLY in a -DCCDM_EXPERIMENTS build
// (CCDM_EXPERIMENTS=1 python -c "from ccdm_stochastic_segmentation_amd import hip; hip.build()"); the shipped library reads no
// environment, allocates nothing and keeps no global state besides the thread-local error string.
#ifdef CCDM_EXPERIMENTS
static inline int exp_env(const char* name) { const char* v = getenv(name); return v ? atoi(v) : 0; }
#else
static inline constexpr int exp_env(const char*) { return 0; }
#endif

// A kernel whose argument block spans several 64-byte lines fetches them where the compiler first needs a field — one scalar-cache
// miss (a round trip to L2: hundreds of cycles) after another, at the top of a block whose dependent chain is the launch time
// (few-pixel stages: one block per CU, every block misses).  This touches every line of a BYTES-long kernarg segment at once and
// waits once: the misses overlap, the compiler's own loads hit.
#ifdef __HIPCC__
template <int BYTES>
__device__ __forceinline__ void warm_kernargs() {
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    static_assert(BYTES <= 512, "eight lines");
    constexpr int L = (BYTES + 63) / 64;                 // lines of the explicit arguments; a line index beyond them re-reads line 0
    unsigned d0, d1, d2, d3, d4, d5, d6, d7;
    // ONE statement: requests and the wait together, so that no destination register is handed out again while its load is in flight
    asm volatile("s_load_dword %0, %8, %9\n\ts_load_dword %1, %8, %10\n\ts_load_dword %2, %8, %11\n\ts_load_dword %3, %8, %12\n\t"
                 "s_load_dword %4, %8, %13\n\ts_load_dword %5, %8, %14\n\ts_load_dword %6, %8, %15\n\ts_load_dword %7, %8, %16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7)
                 : "s"(kp), "i"(0), "i"(L > 1 ? 0x40 : 0), "i"(L > 2 ? 0x80 : 0), "i"(L > 3 ? 0xc0 : 0), "i"(L > 4 ? 0x100 : 0),
                   "i"(L > 5 ? 0x140 : 0), "i"(L > 6 ? 0x180 : 0), "i"(L > 7 ? 0x1c0 : 0));
}
#endif

// ---- conv launch geometry (shared by the kernel dispatcher, the packer and ccdm_conv_slices) ----
struct ConvGeo {
    int TH, TW, waves, MI;   // output tile, waves per block, 32-pixel sub-tiles per wave
};
// up2: the sub-pixel upsample form tiles the low-resolution INPUT space; its blocks carry four phases of accumulators, so it
// never takes the two-sub-tile 8x32 geometry
static inline ConvGeo conv_geo(int Hout, int Wout, int stride, bool up2 = false, bool fine = false) {
    if (up2) return Wout >= 16 ? ConvGeo{8, 16, 4, 1} : ConvGeo{8, 8, 2, 1};
    if (stride == 2) return Wout >= 16 ? ConvGeo{8, 16, 4, 1} : ConvGeo{8, 8, 2, 1};     // (8x16: 4-wave blocks, 57 -> 49 us at 128x128 -> 64x64)
    // by image area, never by batch size (the statistics slices follow the tiling): wide tiles amortise the halo where there are
    // pixels to fill the chip with; the few-pixel levels of a deep U-Net (16x32 and 8x16 at Cityscapes sizes, like LIDC's 16x16 and 8x8)
    // take narrow tiles — more blocks, wider channel chunks, the 3-way tap split at 8x8
    // (measured: C5 shard 15.04 -> 14.63 ms, C4 7.54 -> 7.31 ms per step against the by-width rule; LIDC sizes keep their tiles, and a
    //  32x32 image on 8x16 tiles costs the LIDC step 2.5 %)
    const int area = Hout * Wout;
    // latency slicing (ccdm_conv_args.fine_slices — the one mode whose choices may assume a small batch): 32x32 images on 8x16 tiles,
    // 16x16 on 8x8 tiles with the tap split (batch 8: 1.56 -> 1.48 ms per LIDC denoise step; both lose at batch 64)
    if (fine && area <= 1024) return Wout >= 16 && area > 256 ? ConvGeo{8, 16, 4, 1} : ConvGeo{8, 8, 2, 1};
    if (Wout >= 32 && area > 512) return {8, 32, 4, 2};
    if (Wout >= 16 && area > 128) return {8, 16, 4, 1};
    return {8, 8, 2, 1};
}
// number of 32-wide output-channel tiles one block computes, and the padded tile count
static inline void conv_ntiles(int Cout, int* ntiles_padded, int* NI) {
    int raw = cdiv(Cout, 32);
    if (raw <= 4) { *NI = raw; *ntiles_padded = raw; return; }
    for (int ni = 4; ni >= 2; --ni)
        if (raw % ni == 0) { *NI = ni; *ntiles_padded = raw; return; }
    *NI = 4; *ntiles_padded = cdiv(raw, 4) * 4;
}
static inline int conv_cin_pad(int Cin) { return cdiv(Cin, 32) * 32; }

int launch_conv(const ccdm_conv_args& a, hipStream_t s);
int launch_conv_input_absmax(const ccdm_conv_args& a, float* out, hipStream_t s);
int launch_attention(const float* qkv, float* out, int N, int T, int Ta, int C, int heads, int order, hipStream_t s);
int launch_posterior(const ccdm_post_args& a, hipStream_t s);
int launch_stats_fold(const double* in, int N, int S_in, int C, int S_out, double* out, hipStream_t s);
int launch_attn_block(const ccdm_attn_block_args& a, hipStream_t s);
int launch_resample(const ccdm_resample_args& a, hipStream_t s);
int launch_stem(const ccdm_stem_args& a, hipStream_t s);
int launch_head(const ccdm_head_args& a, const ccdm_post_args& post, hipStream_t s);
bool attn_block_supported(int T, int C, int heads);

}  // namespace ccdm
