// ea_calib.hip -- known byte counts against the L2's memory-side request counters (TCC_EA0_RDREQ[_32B|_64B|_128B],
// TCC_EA0_WRREQ[_64B], TCC_EA0_ATOMIC) in the access shapes the guidance step uses.  Buffers are 1 GiB (four times the
// 256 MiB Infinity Cache) and are touched once, so every byte comes from / goes to HBM.
//   k_read16   : 16 bytes per lane, fully coalesced streaming read         (N bytes)
//   k_read4    :  4 bytes per lane, coalesced                              (N bytes)
//   k_read12   : 12 bytes per lane, contiguous (global_load_dwordx3)       (N bytes)
//   k_gather12 : 12 bytes per lane at a pseudo-random 12-byte record       (M records: 12 M bytes asked for, whole lines moved)
//   k_write16  : 16 bytes per lane streaming store                         (N bytes)
//   k_atomic8  : one 64-bit agent-scope atomicMax per lane on pseudo-random words of a 256 MiB plane (M atomics)
// hipcc --offload-arch=gfx950 -O3 ea_calib.hip -o ea_calib ; rocprofv3 --pmc <counters> -- ./ea_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_read16(const float4* p, size_t n, float* sink) {
    float a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 123.456f) *sink = a;
}
__global__ void k_read4(const float* p, size_t n, float* sink) {
    float a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a == 123.456f) *sink = a;
}
struct __attribute__((packed, aligned(4))) f3 { float x, y, z; };
__global__ void k_read12(const f3* p, size_t n, float* sink) {
    float a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const f3 v = p[i]; a += v.x + v.y + v.z; }
    if (a == 123.456f) *sink = a;
}
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void k_gather12(const f3* p, size_t nrec, size_t m, float* sink) {
    float a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) { const f3 v = p[mix(i) % nrec]; a += v.x + v.y + v.z; }
    if (a == 123.456f) *sink = a;
}
__global__ void k_write16(float4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = float4{1.f, 2.f, 3.f, (float)i};
}
__global__ void k_atomic8(unsigned long long* p, size_t nwords, size_t m) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) atomicMax(&p[mix(i) % nwords], (unsigned long long)i);
}

int main() {
    const size_t N = 1ull << 30;
    void *a, *b; float* sink;
    CK(hipMalloc(&a, N)); CK(hipMalloc(&b, N)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 0, N)); CK(hipMemset(b, 0, N)); CK(hipDeviceSynchronize());
    const int G = 4096, T = 256;
    const size_t M = 1ull << 22;   // gathers / atomics
    // alternate the buffers so that nothing a kernel reads is left in the Infinity Cache by its predecessor (1 GiB each)
    hipLaunchKernelGGL(k_read16, dim3(G), dim3(T), 0, 0, (const float4*)a, N / 16, sink);
    hipLaunchKernelGGL(k_read4, dim3(G), dim3(T), 0, 0, (const float*)b, N / 4, sink);
    hipLaunchKernelGGL(k_read12, dim3(G), dim3(T), 0, 0, (const f3*)a, N / 12, sink);
    hipLaunchKernelGGL(k_read16, dim3(G), dim3(T), 0, 0, (const float4*)b, N / 16, sink);     // flush a out of the caches
    hipLaunchKernelGGL(k_gather12, dim3(G), dim3(T), 0, 0, (const f3*)a, N / 12, M, sink);
    hipLaunchKernelGGL(k_write16, dim3(G), dim3(T), 0, 0, (float4*)b, N / 16);
    hipLaunchKernelGGL(k_read16, dim3(G), dim3(T), 0, 0, (const float4*)a, N / 16, sink);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_atomic8, dim3(G), dim3(T), 0, 0, (unsigned long long*)b, (size_t)(256u << 20) / 8, M);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_atomic8: %zu agent-scope 64-bit atomics on random words in %.3f ms = %.1f atomics/ns\n", M, ms, M / (ms * 1e6));
    // ... and on a plane the size of a z-key plane (2 MB: 512 x 512 x 8 bytes), the step's case
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_atomic8, dim3(G), dim3(T), 0, 0, (unsigned long long*)b, (size_t)(2u << 20) / 8, M);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_atomic8 on a 2 MB plane: %.3f ms = %.1f atomics/ns\n", ms, M / (ms * 1e6));
    printf("bytes: read16/read4/write16 %zu, read12 %zu, gather12 %zu records (%zu bytes asked), atomic8 %zu ops\n", N, N / 12 * 12, M, M * 12, M);
    return 0;
}
