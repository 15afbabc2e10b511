// What does ds_read_b64_tr_b16 return?  LDS holds halfs whose value = their index; every lane passes its own address.
//   pattern 0: lane L reads from byte 8*L (a contiguous 64-lane x 4-half image)
//   pattern 1: lane L (group-local l = L & 15, group g = L >> 4) reads row (l >> 2) of a [rows][16] block at column 4*(l & 3),
//              row stride 64 B, group g offset 1024 B
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int pattern) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)(float)(i % 2048);
    __syncthreads();
    const int L = threadIdx.x;
    unsigned addr;
    if (pattern == 0) addr = 8 * L;
    else { const int l = L & 15, g = L >> 4; addr = g * 1024 + (l >> 2) * 64 + (l & 3) * 8; }
    addr += (unsigned)(size_t)lds;     // LDS byte address
    f16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[L * 4 + j] = (float)v[j];
}
int main() {
    float* d; (void)hipMalloc(&d, 64 * 4 * 4);
    float h[256];
    for (int p = 0; p < 2; ++p) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, p);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d (half index each lane received: lane: e0 e1 e2 e3)\n", p);
        for (int L = 0; L < 64; ++L) printf("%2d: %4.0f %4.0f %4.0f %4.0f%s", L, h[L * 4], h[L * 4 + 1], h[L * 4 + 2], h[L * 4 + 3], (L % 4 == 3) ? "\n" : "   ");
    }
    return 0;
}
