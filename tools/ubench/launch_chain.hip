// What does a dependent kernel boundary cost on this box?  N trivial kernels back to back on one stream — launched eagerly and as one
// captured HIP graph — for a 1-block grid and a 256- / 768-block grid of 256 threads, and with each kernel leaving `dirty` KB per block
// of freshly written lines behind (the write-back of what a kernel leaves dirty in L2 sits at the boundary).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_chain.hip -o tools/ubench/launch_chain && tools/ubench/launch_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_touch(float* p, int words_per_block, int iter) {
    // every block rewrites its own slab (words_per_block floats), coalesced
    float* q = p + (size_t)blockIdx.x * words_per_block;
    for (int i = threadIdx.x; i < words_per_block; i += blockDim.x) q[i] = (float)(iter + i);
}

static double run(hipStream_t s, bool graph, int nk, int blocks, int words, float* buf, int reps) {
    hipGraph_t g = nullptr;
    hipGraphExec_t ex = nullptr;
    if (graph) {
        hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, s, buf, words, i);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        hipGraphLaunch(ex, s);
    } else {
        for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, s, buf, words, i);
    }
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        if (graph) hipGraphLaunch(ex, s);
        else for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, s, buf, words, i);
    }
    hipStreamSynchronize(s);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (ex) hipGraphExecDestroy(ex);
    if (g) hipGraphDestroy(g);
    return us / (double)(reps * nk);
}

int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    float* buf;
    const size_t bytes = (size_t)768 * 256 * 1024;      // 768 blocks x up to 256 KB
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    const int nk = 88, reps = 40;
    printf("us per kernel, %d dependent kernels per pass (the LIDC denoise step has 88)\n", nk);
    for (int blocks : {1, 256, 768}) {
        for (int kb : {0, 4, 32, 128}) {
            const int words = kb == 0 ? 64 : kb * 256;
            const double e = run(s, false, nk, blocks, words, buf, reps), g = run(s, true, nk, blocks, words, buf, reps);
            printf("blocks %4d  %4d KB written per block (%7.1f MB per kernel): eager %6.2f  graph %6.2f\n", blocks, kb,
                   (double)blocks * words * 4 / 1e6, e, g);
        }
    }
    return 0;
}
