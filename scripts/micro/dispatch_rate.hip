// Workgroup launch-rate probe for gfx950: how many workgroups per microsecond the dispatchers start, as a function of
// workgroup size, LDS allocation and kernel-argument size.  Build: hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big { int pad[160]; };  // 640-byte by-value argument like the step's Ctx

template <int LDS_BYTES>
__global__ void k_probe(int* out, int never) {
    __shared__ int s[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    if ((int)blockIdx.x == never) {  // keeps the allocation alive, never taken
        s[threadIdx.x] = never;
        __syncthreads();
        out[threadIdx.x] = s[(threadIdx.x + 1) % 64];
    }
}
template <int LDS_BYTES>
__global__ void k_probe_big(Big a, int* out, int never) {
    __shared__ int s[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    if ((int)blockIdx.x == never) {
        s[threadIdx.x] = a.pad[threadIdx.x % 160];
        __syncthreads();
        out[threadIdx.x] = s[(threadIdx.x + 1) % 64];
    }
}
// a workgroup that lives ~work iterations of dependent ALU (no memory): residency effects
template <int LDS_BYTES>
__global__ void k_probe_busy(int* out, int never, int work) {
    __shared__ int s[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    float x = threadIdx.x;
    for (int i = 0; i < work; i++) x = x * 1.0001f + 0.5f;
    if ((int)blockIdx.x == never || x == 12345.678f) {
        s[threadIdx.x] = never;
        __syncthreads();
        out[threadIdx.x] = s[(threadIdx.x + 1) % 64];
    }
}

template <typename F>
static float time_us(F launch, int reps = 20) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main() {
    int* out;
    hipMalloc(&out, 4096);
    Big big{};
    printf("%-28s %8s %8s %10s %10s\n", "kernel", "threads", "WGs", "us", "WG/us");
    const int grids[] = {256, 1024, 4096, 16384};
    for (int tpb : {64, 256, 1024})
        for (int g : grids) {
            float t0 = time_us([&] { hipLaunchKernelGGL(k_probe<0>, dim3(g), dim3(tpb), 0, 0, out, -1); });
            float t1 = time_us([&] { hipLaunchKernelGGL(k_probe<20480>, dim3(g), dim3(tpb), 0, 0, out, -1); });
            float t2 = time_us([&] { hipLaunchKernelGGL(k_probe<65536>, dim3(g), dim3(tpb), 0, 0, out, -1); });
            float t3 = time_us([&] { hipLaunchKernelGGL(k_probe_big<20480>, dim3(g), dim3(tpb), 0, 0, big, out, -1); });
            printf("%-28s %8d %8d %10.2f %10.1f\n", "lds0", tpb, g, t0, g / t0);
            printf("%-28s %8d %8d %10.2f %10.1f\n", "lds20k", tpb, g, t1, g / t1);
            printf("%-28s %8d %8d %10.2f %10.1f\n", "lds64k", tpb, g, t2, g / t2);
            printf("%-28s %8d %8d %10.2f %10.1f\n", "lds20k+640B args", tpb, g, t3, g / t3);
        }
    for (int work : {200, 1000, 4000})
        for (int g : {1024, 4096, 16384}) {
            float t = time_us([&] { hipLaunchKernelGGL(k_probe_busy<20480>, dim3(g), dim3(256), 0, 0, out, -1, work); });
            printf("busy%-5d lds20k %16d %8d %10.2f %10.1f\n", work, 256, g, t, g / t);
        }
    // 2-D grid like the step's (x = role blocks, y = image)
    for (int by : {1, 16}) {
        float t = time_us([&] { hipLaunchKernelGGL(k_probe<20480>, dim3(4096 / by, by), dim3(256), 0, 0, out, -1); });
        printf("lds20k grid (%d,%d) %23d %10.2f %10.1f\n", 4096 / by, by, 4096, t, 4096 / t);
    }
    return 0;
}
