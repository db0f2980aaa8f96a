// mfma_probe.hip -- fragment layouts of v_mfma_f32_32x32x16_f16 on gfx950, checked against the layouts the geometry
// decoder's kernels (csrc/k_geo.inc) assume, and the lane-linear destination of global_load_lds_dwordx4.
//   hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe && ./mfma_probe
// Hypotheses (A is M x K, B is K x N, C is M x N, lane l, hi = l >> 5):
//   A: lane l holds A[l & 31][8 hi + j], j = 0..7         B: lane l holds B[8 hi + j][l & 31]
//   C: lane l, reg r holds C[(r & 3) + 8 (r >> 2) + 4 hi][l & 31]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_mfma(const _Float16* A, const _Float16* B, float* C) {   // A [32][16], B [16][32] row-major, C [32][32]
    const int l = threadIdx.x, hi = l >> 5;
    half8 a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = A[(l & 31) * 16 + 8 * hi + j];
        b[j] = B[(8 * hi + j) * 32 + (l & 31)];
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + (l & 31)] = acc[r];
}

__global__ void k_glds(const uint4* g, uint4* out) {   // 256 threads: wave w copies 64 x 16 B to LDS at base w KiB, lane-linear
    __shared__ uint4 lds[256];
    const int w = threadIdx.x >> 6;
    // lane i asks for source element (its wave's 64) in REVERSED order; the destination must still be lane-linear
    const uint4* src = g + w * 64 + (63 - (threadIdx.x & 63));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + w * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}

int main() {
    std::vector<_Float16> A(32 * 16), B(16 * 32);
    srand(1);
    for (auto& v : A) v = (_Float16)(float)(rand() % 7 - 3);
    for (auto& v : B) v = (_Float16)(float)(rand() % 5 - 2);
    _Float16 *dA, *dB; float* dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(32 * 32);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++)
        for (int n = 0; n < 32; n++) {
            float ref = 0;
            for (int k = 0; k < 16; k++) ref += (float)A[i * 16 + k] * (float)B[k * 32 + n];
            bad += ref != C[i * 32 + n];
        }
    printf("mfma_f32_32x32x16_f16 layout hypothesis: %s (%d of 1024 wrong)\n", bad ? "WRONG" : "ok", bad);

    std::vector<unsigned> g(256 * 4);
    for (int i = 0; i < 256 * 4; i++) g[i] = i;
    uint4 *dg, *dout;
    hipMalloc(&dg, 4096); hipMalloc(&dout, 4096);
    hipMemcpy(dg, g.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_glds, dim3(1), dim3(256), 0, 0, dg, dout);
    std::vector<unsigned> o(256 * 4);
    hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
    int badg = 0;
    for (int t = 0; t < 256; t++) {
        const int w = t >> 6, srcel = w * 64 + (63 - (t & 63));
        for (int q = 0; q < 4; q++) badg += o[t * 4 + q] != (unsigned)(srcel * 4 + q);
    }
    printf("global_load_lds_dwordx4: destination = wave base + 16 * lane, source per lane: %s (%d words wrong)\n", badg ? "WRONG" : "ok", badg);
    return (bad || badg) ? 1 : 0;
}
