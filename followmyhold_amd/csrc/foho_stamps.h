// foho_stamps.h -- ALL development instrumentation of the step's kernels, in one place.  The product library (plain `make`)
// compiles every macro below to nothing / a constant the optimiser folds; `make STAMPS=1` builds libfoho_hip_stamps.so
// (never shipped, scripts/dev_stamps.py) with in-kernel time stamps, per-workgroup spans and ABLATION switches whose
// results are invalid by design.  The kernels and host.inc use only these names -- no `#ifdef` in the product sources:
//   DBG(i) / DBGW(i)              100 MHz wall clock of thread 0 into slot i (after draining the wait counters)
//   KSPAN(k)                      start / end stamp of every workgroup of launch k
//   FOHO_ABLATED(cfg, bit)        is ablation `bit` of cfg.dbg switched on?  (false in the product build)
//   FOHO_ABLATE_RETURN(cfg, bit)  leave the role here when it is
//   FOHO_UNLESS_ABLATED(cfg, bit) prefix of a statement that the ablation removes
//   FOHO_DEV_ENV("NAME")          getenv() of a development knob (nullptr in the product build)
#pragma once

#ifdef FOHO_STAMPS
__device__ unsigned long long g_dbg[1024];
#define DBG(i)                                                  \
    do {                                                        \
        if (threadIdx.x == 0) g_dbg[i] = wall_clock64();        \
    } while (0)
#define DBGW(i)                                  \
    do {                                         \
        __builtin_amdgcn_s_waitcnt(0);           \
        DBG(i);                                  \
    } while (0)
extern "C" void foho_debug_clear(void) {
    static unsigned long long z[1024];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z));
}
extern "C" void foho_debug_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), 1024 * 8); }
// KSPAN(k): device-side span of a launch -- every workgroup stores its own start / end stamp (plain stores, own slots);
// foho_debug_spans() copies the table: [kernel][workgroup][2], KS_WG workgroups per kernel
constexpr int KS_K = 6, KS_WG = 8192;
__device__ unsigned long long g_span[KS_K * KS_WG * 2];
struct KSpan {
    unsigned long long* p;
    __device__ __forceinline__ explicit KSpan(int k) {
        const unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        p = &g_span[((size_t)k * KS_WG + (lin < KS_WG ? lin : KS_WG - 1)) * 2];
        if (threadIdx.x == 0) p[0] = wall_clock64();
    }
    __device__ __forceinline__ ~KSpan() {
        if (threadIdx.x == 0) p[1] = wall_clock64();
    }
};
extern "C" void foho_debug_spans(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(g_span)); }
extern "C" void foho_debug_spans_clear(void) { (void)hipMemset((void*)nullptr, 0, 0); void* d = nullptr; (void)hipGetSymbolAddress(&d, HIP_SYMBOL(g_span)); (void)hipMemset(d, 0, sizeof(g_span)); }
#define KSPAN(k) KSpan kspan_(k)
#define FOHO_ABLATED(cfg, bit) (((cfg).dbg & (bit)) != 0)
#define FOHO_ABLATED_WINDOW true
#define DBG_VALUE(i, v) (g_dbg[i] = (v))
#define FOHO_DEV_ENV(name) getenv(name)
#else
#define DBG(i) \
    do {       \
    } while (0)
#define DBGW(i) \
    do {        \
    } while (0)
#define KSPAN(k) \
    do {         \
    } while (0)
#define FOHO_ABLATED(cfg, bit) false
#define FOHO_ABLATED_WINDOW false
#define DBG_VALUE(i, v) ((void)0)
#define FOHO_DEV_ENV(name) ((const char*)nullptr)
#endif
#define FOHO_ABLATE_RETURN(cfg, bit)        \
    do {                                    \
        if (FOHO_ABLATED(cfg, bit)) return; \
    } while (0)
#define FOHO_UNLESS_ABLATED(cfg, bit) if (!FOHO_ABLATED(cfg, bit))

