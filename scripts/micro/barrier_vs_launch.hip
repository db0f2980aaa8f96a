// Micro-benchmark: a chain of N dependent phases, each touching a small array (one load + one store per thread),
//   (a) as N kernel launches replayed from a hipGraph,
//   (b) as ONE persistent kernel with N grid barriers (16 arrival counters on separate cache lines, release / acquire
//       fences at agent scope, pollers with s_sleep).
// Answers what a phase boundary costs on this part either way.  hipcc --offload-arch=gfx950 -O3 barrier_vs_launch.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_phase(const float* in, float* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[(i * 97 + 13) % n] + 1.0f;   // reads what another workgroup of the previous phase wrote
}

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned phase, unsigned nwg) {
    __syncthreads();
    const unsigned lin = blockIdx.x;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        atomicAdd(&bar[(lin & 15) * 32], 1u);
    }
    if (threadIdx.x < 16) {
        const unsigned want = phase * ((nwg + 15 - threadIdx.x) / 16);
        while (__hip_atomic_load(&bar[threadIdx.x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(256) void k_persistent(float* a, float* b, int n, int nphase, unsigned* bar) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float* in = a;
    float* out = b;
    for (int p = 0; p < nphase; p++) {
        if (i < n) out[i] = in[(i * 97 + 13) % n] + 1.0f;
        grid_barrier(bar, (unsigned)(p + 1), gridDim.x);
        float* t = in; in = out; out = t;
    }
}

int main(int argc, char** argv) {
    const int nphase = 60;
    for (int nwg : {64, 128, 256, 512, 1024}) {
        const int n = nwg * 256;
        float *a, *b;
        unsigned* bar;
        CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&bar, 16 * 32 * 4));
        CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
        hipStream_t s; CK(hipStreamCreate(&s));
        // (a) graph of launches
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int p = 0; p < nphase; p++) hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(256), 0, s, (p & 1) ? b : a, (p & 1) ? a : b, n);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 20; r++) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms_g; CK(hipEventElapsedTime(&ms_g, e0, e1));
        // (b) persistent kernel
        float ms_p = 0;
        for (int r = 0; r < 23; r++) {
            CK(hipMemsetAsync(bar, 0, 16 * 32 * 4, s));
            if (r == 3) CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k_persistent, dim3(nwg), dim3(256), 0, s, a, b, n, nphase, bar);
        }
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms_p, e0, e1));
        std::vector<float> h(n);
        CK(hipMemcpy(h.data(), (nphase & 1) ? b : a, n * 4, hipMemcpyDeviceToHost));
        printf("%5d workgroups: graph of launches %.2f us per phase | persistent kernel + grid barrier %.2f us per phase (incl. launch + memset / %d phases)  check %.0f\n",
               nwg, ms_g * 1e3 / (20 * nphase), ms_p * 1e3 / (20 * nphase), nphase, h[5]);
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(bar));
    }
    return 0;
}
