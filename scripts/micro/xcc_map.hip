// xcc_map.hip -- which XCD a workgroup of a 2-D grid lands on: XCC_ID per (blockIdx.x, blockIdx.y) against the linear id
// blockIdx.x + gridDim.x * blockIdx.y modulo 8.  hipcc --offload-arch=gfx950 -O3 xcc_map.hip -o xcc_map && ./xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x + gridDim.x * blockIdx.y] = (int)(x & 0xf);
}
int main() {
    for (int gx : {1019, 1359, 64}) {
        const int gy = 5;
        int* d; hipMalloc(&d, gx * gy * 4);
        hipLaunchKernelGGL(k, dim3(gx, gy), dim3(256), 0, 0, d);
        std::vector<int> h(gx * gy); hipMemcpy(h.data(), d, gx * gy * 4, hipMemcpyDeviceToHost);
        int agree = 0, first = h[0];
        for (int i = 0; i < gx * gy; i++) agree += (h[i] == (first + i) % 8);
        printf("grid %d x %d: xcc of block 0 = %d, blocks with xcc == (xcc0 + linear) %% 8: %d of %d; first 20:", gx, gy, first, agree, gx * gy);
        for (int i = 0; i < 20; i++) printf(" %d", h[i]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
