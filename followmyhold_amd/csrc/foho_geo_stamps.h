// Development instrumentation of foho_geo.hip (the counterpart of foho_stamps.h for the step's kernels): the product build defines
// none of the switches below and every hook is empty.
#pragma once
// The phased GEMM (k_geo_gemm8p).  -DP8_STAMPS: per-wave sums of the K loop's segment durations (shader clocks),
// workgroup 0 -> foho_geo_p8_stamps() (scripts/dev_p8_stamps.py; ~40 cycles per stamp, and the stamp's wait retires LDS reads early).
// -DP8_TIMELINE: per tile, stamps of {entry, first matrix instruction, K loop end, wave groups re-joined, exit} + HW_ID / XCC_ID ->
// foho_geo_p8_timeline() (scripts/dev_p8_timeline.py).  Without either the hooks are empty.  k_geo_gemm_d4 carries the P8_STAMP hooks too
// (scripts/dev/d4_stamps.py) and a -DD4_FILL_ONLY ablation (scripts/dev/d4_fill.py: the ring's fill without the matrix work).
#ifdef P8_STAMPS
__device__ unsigned long long g_p8[8][8];
extern "C" __attribute__((visibility("default"))) void foho_geo_p8_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p8), sizeof(g_p8)); }
#define P8_STAMP_DECL                                  \
    unsigned long long ts[7];                          \
    int seg[6] = {0, 0, 0, 0, 0, 0};                   \
    P8_STAMP(0);                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define P8_STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[i]))
#define P8_ACC()                                                                                           \
    do {                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < 6; q_++) seg[q_] += (int)(long long)(ts[q_ + 1] - ts[q_]); \
        ts[0] = ts[6];                                                                                     \
    } while (0)
#define P8_STAMP_DUMP(w, nk)                                                                           \
    do {                                                                                               \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                                              \
            for (int q_ = 0; q_ < 6; q_++) g_p8[w][q_] = (unsigned long long)(long long)seg[q_];       \
            g_p8[w][6] = (nk);                                                                         \
        }                                                                                              \
    } while (0)
#else
#define P8_STAMP_DECL do { } while (0)
#define P8_STAMP(i) do { } while (0)
#define P8_ACC() do { } while (0)
#define P8_STAMP_DUMP(w, nk) do { } while (0)
#endif
#ifdef P8_TIMELINE
__device__ unsigned long long g_p8tl[4096][8];
extern "C" __attribute__((visibility("default"))) void foho_geo_p8_timeline(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p8tl), sizeof(g_p8tl)); }
#define P8_TL_DECL unsigned long long tl_[8]
#define P8_TL(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_[i]))
#define P8_TL_DUMP(L)                                                                                                             \
    do {                                                                                                                          \
        if (threadIdx.x == 0 && (L) < 4096) {                                                                                     \
            unsigned hwid_, xcc_;                                                                                                 \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_));                                                   \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                                   \
            for (int i_ = 0; i_ < 5; i_++) g_p8tl[L][i_] = tl_[i_];                                                               \
            g_p8tl[L][5] = hwid_;                                                                                                 \
            g_p8tl[L][6] = xcc_;                                                                                                  \
            g_p8tl[L][7] = ((tl_[5] - tl_[0]) << 32) | ((tl_[6] - tl_[5]) & 0xffffffffull); /* entry -> DMA issued | -> landed */ \
        }                                                                                                                         \
    } while (0)
#else
#define P8_TL_DECL do { } while (0)
#define P8_TL(i) do { } while (0)
#define P8_TL_DUMP(L) do { } while (0)
#endif
