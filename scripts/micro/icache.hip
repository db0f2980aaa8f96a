// Micro-benchmark: what does code that is executed ONCE per wave cost, and does the instruction cache survive a kernel
// boundary?  k_line<ID> is NI dependent v_fma_f32 with literal constants (8 bytes each: 4000 of them = 32 KB of straight-line
// code), one wave per workgroup, 256 workgroups.  Every wave stamps its own start / end (s_memrealtime, 100 MHz).
//   A A A A       the same 32 KB kernel back to back: if the cache survived the boundary, launches 2.. run warm
//   A B A B       two kernels, 64 KB together (the cache is 64 KB per two CUs)
//   A B C A B C   96 KB: cyclic eviction
//   loop          the same instructions as a 16-iteration loop over 1/16 of the code: warm by construction
// hipcc --offload-arch=gfx950 -O3 icache.hip -o icache
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NI = 4000;

template <int ID, int I>
struct Chain {
    static __device__ __forceinline__ float run(float x, float y) {
        // a literal that differs per instruction: no way to share or loop
        constexpr float k = 1.0f + (float)((I * 7919 + ID * 104729) % 65521) * (1.0f / 4194304.0f);
        x = __builtin_fmaf(x, k, y);
        return Chain<ID, I - 1>::run(x, y);
    }
};
template <int ID>
struct Chain<ID, 0> {
    static __device__ __forceinline__ float run(float x, float) { return x; }
};

template <int ID>
__global__ __launch_bounds__(64) void k_line(float* out, unsigned long long* stamp, int slot) {
    const unsigned long long t0 = wall_clock64();
    float x = (float)threadIdx.x * 1e-3f, y = out[blockIdx.x & 7];
    // 8 blocks of 500 (template recursion depth)
    x = Chain<ID, 500>::run(x, y); x = Chain<ID + 1, 500>::run(x, y); x = Chain<ID + 2, 500>::run(x, y); x = Chain<ID + 3, 500>::run(x, y);
    x = Chain<ID + 4, 500>::run(x, y); x = Chain<ID + 5, 500>::run(x, y); x = Chain<ID + 6, 500>::run(x, y); x = Chain<ID + 7, 500>::run(x, y);
    if (x == 123.456f) out[blockIdx.x] = x;
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        stamp[(slot * 256 + blockIdx.x) * 2] = t0;
        stamp[(slot * 256 + blockIdx.x) * 2 + 1] = t1;
    }
}

__global__ __launch_bounds__(64) void k_loop(float* out, unsigned long long* stamp, int slot) {
    const unsigned long long t0 = wall_clock64();
    float x = (float)threadIdx.x * 1e-3f, y = out[blockIdx.x & 7];
#pragma nounroll
    for (int r = 0; r < 16; r++) x = Chain<99, 250>::run(x, y);
    if (x == 123.456f) out[blockIdx.x] = x;
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        stamp[(slot * 256 + blockIdx.x) * 2] = t0;
        stamp[(slot * 256 + blockIdx.x) * 2 + 1] = t1;
    }
}

int main() {
    float* out;
    unsigned long long* stamp;
    const int NS = 64;
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(out, 0, 4096));
    CK(hipMalloc(&stamp, NS * 256 * 16));
    CK(hipMemset(stamp, 0, NS * 256 * 16));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const char* seq = "AAAA.ABAB.ABCABC.LLL.ALAL";
    std::vector<char> names;
    int slot = 0;
    for (int rep = 0; rep < 2; rep++) {  // second repetition: everything has been loaded once (code object, page tables)
        slot = 0;
        names.clear();
        for (const char* p = seq; *p; p++) {
            if (*p == '.') { CK(hipStreamSynchronize(s)); continue; }
            if (*p == 'A') hipLaunchKernelGGL(k_line<0>, dim3(256), dim3(64), 0, s, out, stamp, slot);
            if (*p == 'B') hipLaunchKernelGGL(k_line<20>, dim3(256), dim3(64), 0, s, out, stamp, slot);
            if (*p == 'C') hipLaunchKernelGGL(k_line<40>, dim3(256), dim3(64), 0, s, out, stamp, slot);
            if (*p == 'L') hipLaunchKernelGGL(k_loop, dim3(256), dim3(64), 0, s, out, stamp, slot);
            names.push_back(*p);
            slot++;
        }
        CK(hipStreamSynchronize(s));
    }
    std::vector<unsigned long long> h(NS * 256 * 2);
    CK(hipMemcpy(h.data(), stamp, h.size() * 8, hipMemcpyDeviceToHost));
    printf("%d dependent v_fma_f32 per wave, one wave per workgroup, 256 workgroups; wave life in us (100 MHz stamps)\n", NI);
    for (int k = 0; k < slot; k++) {
        std::vector<double> life;
        for (int w = 0; w < 256; w++) life.push_back((double)(h[(k * 256 + w) * 2 + 1] - h[(k * 256 + w) * 2]) / 100.0);
        std::sort(life.begin(), life.end());
        printf("launch %2d  %c   min %.2f  median %.2f  max %.2f\n", k, names[k], life[0], life[128], life[255]);
    }
    return 0;
}
