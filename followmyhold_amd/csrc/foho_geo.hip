// foho_geo.hip -- the ShapeVAE geometry decoder of `latent2sdf` on the MI355X matrix cores.
//
// Reference: third_party_patches/hy3dgen/shapegen/pipelines.py:292-313 -- every inner iteration of phases B and C decodes
// the current clean-sample estimate into a 65^3 occupancy grid: 35 chunks x 8000 queries through `vae.geo_decoder`
// (Fourier embedding -> query projection -> ONE cross-attention block over the 3072 x 1024 latent tokens, 16 heads of
// 64 -> MLP 1024 -> 4096 -> 1024 -> LayerNorm -> logit).  33.7 MFLOP per query, 9.25 TFLOP per grid: the one place on the
// path where the matrix cores decide the time.
//
// Design for gfx950:
//   * fp16 storage, fp32 accumulation on v_mfma_f32_32x32x16_f16 (the reference runs the VAE in fp16, PL:522).
//   * the 8000-query chunks are gone: one call decodes all queries, in row blocks sized so that the block's activations
//     (14.5 KB per query) stay inside the 256 MB Infinity Cache between the kernels of the chain.
//   * K and V of the latent tokens are projected once per call; V is kept TRANSPOSED and key-permuted so that both
//     operands of P.V are 16-byte contiguous fragments (no LDS transpose, no cross-lane exchange of P).
//   * attention: swapped QK^T (S^T = K Q^T) puts a query's scores in ONE lane column -> row max / row sum are in-lane
//     v_max3 / adds plus one exchange between the two half-waves; softmax scale and log2(e) are folded into Q;
//     the running max is only raised when a tile exceeds it by more than 2^6 (deferred rescale).
//   * GEMMs: 128 x 128 x 64 tiles, 4 waves x (64 x 64), operands staged by LDS-DMA (global_load_lds_dwordx4; tile t+1 in
//     flight while tile t is multiplied) into a double-buffered, XOR-swizzled image (swizzle applied on the SOURCE
//     address, the DMA's destination is lane-linear), conflict-free ds_read_b128 fragments requested one k-step ahead of
//     the MFMAs that use them (inline asm with hand-counted lgkmcnt: see the kernel); bias / GELU / softmax scale / residual fused into the epilogue, which goes
//     through LDS so that global stores are whole 128-byte row segments.
//   * XCD-aware block order everywhere (workgroup L runs on XCD L mod 8; K/V of two heads or the weight panel of a GEMM
//     stay in that XCD's 4 MB L2).
//
// Translation unit of its own: compiled WITHOUT -ffp-contract=off / correctly rounded division (nothing here decides a
// face index), linked into libfoho_hip.so next to foho_step.hip.
#include <hip/hip_runtime.h>
#include <string>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <algorithm>
#include <stdlib.h>

#include "../../include/foho_hip.h"
#include "foho_geo_stamps.h"

namespace geo {

typedef _Float16 h16;
typedef h16 half8 __attribute__((ext_vector_type(8)));
typedef h16 half4 __attribute__((ext_vector_type(4)));
typedef h16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

thread_local char g_err[256] = "";
static int fail(int code, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}
static int fail(int code, const std::string& what) { return fail(code, what.c_str()); }

// 16-byte chunk `c` of row `row` of a [rows][64 halfs] LDS tile (128-byte rows) lives in slot c ^ swz(row): the 32 rows x
// 2 chunks a 32x32x16 fragment read touches then fall on 16 distinct 16-byte bank groups within each of ds_read_b128's
// lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (and their upper-half twins) -- conflict-free.
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// GEMM  C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N]) (+ R[M,N]),  fp16 in / fp32 accumulate / fp16 out.
// N % 128 == 0, K % 64 == 0, all leading dimensions multiples of 8 halfs (16-byte rows).
// ------------------------------------------------------------------------------------------------
constexpr int GM = 128, GN = 128, GK = 64;
constexpr int EP_GELU = 1, EP_RESID = 2;
constexpr int CPAD = 72;  // halfs per row of the epilogue's LDS image (64 + 8: keeps 16-byte alignment, spreads banks)

// GELU in torch's default (erf) form.  erf by Abramowitz-Stegun 7.1.26 -- one v_exp_f32, one v_rcp_f32, five fmas; absolute
// error <= 1.5e-7 + the 1-ulp primitives, three orders below the fp16 rounding of the result (erff() costs ~3x as much
// and was a third of the fc1 GEMM's time in the first version of this kernel).
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);   // erf(|v| / sqrt 2)
    return 0.5f * v + 0.5f * fabsf(v) * e;                                                   // v erf(v / sqrt 2) = |v| erf(|v| / sqrt 2)
}

// GELU of two values at once on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: one instruction per PAIR) -- the fc1 epilogue is
// bound by exactly this vector work (a 256 x 256 tile's 65 536 GELUs with nothing to overlap them at one workgroup per CU) ...
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
    // ... and with ONE transcendental per value instead of two (the reciprocal and the exponential of 7.1.26 were half of the
    // epilogue's issue time: quarter-rate instructions): Abramowitz-Stegun 7.1.28, erf(x) = 1 - (1 + a1 x + ... + a6 x^6)^-16 for
    // x >= 0, |error| <= 3e-7 (1.8e-6 after the four squarings in fp32) -- gelu(v) = max(v, 0) - |v| / 2 (1 + ...)^-16, absolute
    // error <= 8e-7: three orders below the fp16 rounding of the result, like 7.1.26's.  A huge |v| overflows the power to +inf,
    // whose reciprocal is the right 0.
    f32x2 av, mx;
    av[0] = fabsf(v[0]), av[1] = fabsf(v[1]);
    mx[0] = fmaxf(v[0], 0.0f), mx[1] = fmaxf(v[1], 0.0f);
    const f32x2 x = av * 0.70710678118654752f;
    f32x2 p = x * 0.0000430638f + 0.0002765672f;
    p = p * x + 0.0001520143f;
    p = p * x + 0.0092705272f;
    p = p * x + 0.0422820123f;
    p = p * x + 0.0705230784f;
    f32x2 s = p * x + 1.0f;
    s = s * s;
    s = s * s;
    s = s * s;
    s = s * s;
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(s[0]), r[1] = __builtin_amdgcn_rcpf(s[1]);
    return (av * -0.5f) * r + mx;
}

// Epilogue of both GEMM kernels for one 64 (M) x 64 (N) part of a wave's tile, accumulators t[n tile][m tile] in the transposed
// C layout (D rows = n, D columns = m: a lane holds, for ONE m, 4 consecutive n per register group).  bias / GELU / scale in
// fp32, rounded to fp16, through a wave-private LDS image (64 rows x CPAD halfs) to whole-row 16-byte stores; at read-back
// the residual is added (EP_RESID) or the value is multiplied by gelu'(Z) (EP_GELUBWD: the backward of the MLP's
// activation, Z = the saved pre-activation, passed in R).  EP_SAVEZ: the pre-activation itself goes to C2 first (same shape as
// C).  EP_TRANS: a TRANSPOSED copy goes to C2[n][pos(m)], pos = m with bits 2 and 3 of (m & 15) swapped -- the order in
// which an MFMA operand built from a C-layout accumulator holds its k index (see the attention kernels): the backward
// attention kernel reads Q^T and dO^T from such copies.  m0 / n0: first row / column of the 64 x 64 part.
constexpr int EP_GELUBWD = 4, EP_SAVEZ = 8, EP_TRANS = 16;
// EP_QNORM (hy3dgen's qk_norm on the query side): the 64 columns of an epilogue part are exactly one head; every row of the
// part is LayerNorm-ed over them (gain / bias / eps: 129 floats passed in the R slot) BEFORE the scale, from the fp16-rounded
// projection, the way the module normalises c_q's output per head.
constexpr int EP_QNORM = 32;
// LayerNorm folded into the GEMMs on either side of it (forward only; chain_latent_side explains the algebra):
// EP_STATS (with EP_RESID): besides the output, the statistics of every output row's 64 columns of this part -- their mean and the sum
//   of squared deviations from it, taken from the fp16-rounded values the output holds -- go to ((float*)C2)[(row * ldc2 + n0 / 64) * 2],
//   ldc2 = N / 64 slots per row; k_geo_rowstat_finish merges a row's slots (exactly: Chan's update, no E[x^2] - mean^2).
// EP_PREAFF (with EP_GELU): the GEMM ran on the UN-normalised rows with gamma folded into the weights; the pre-activation is
//   rstd[m] * acc - (rstd[m] mean[m]) * s[n] + b'[n]: (rstd, rstd * mean) per row as float2 in the R slot, b' = bias[0..N), s = bias[N..2N)
//   (ldr = N).
// EP_LOGIT (with EP_RESID): the output itself is NOT stored; per row and part (mean, sum of squared deviations, sum of value x gw[n])
//   go to ((float*)C2)[(row * ldc2 + n0 / 64) * 4], gw = bias[N..2N) (N = 64 ldc2): what ln_post + output_proj need of the row.
constexpr int EP_STATS = 64, EP_PREAFF = 128, EP_LOGIT = 256;
// EP_QKN (with EP_PREAFF; the fused q | k | v projection of a ShapeVAE transformer layer, foho_vae.inc): the output's columns are
// [Q of all heads | K of all heads | V of all heads] (N = 3 width, ldr = N under EP_PREAFF); the 64 columns of a part in the first
// third are LayerNorm-ed with q_norm's gain / bias / eps, of a part in the second third with k_norm's, the last third is left alone:
// 129 floats each at bias[2 N ..] and bias[2 N + 132 ..] (hy3dgen's qk_norm).  With EP_SAVEZ the un-normalised projection goes to C2.
constexpr int EP_QKN = 512;
// Fused for the ShapeVAE transformer (round 6, foho_vae.inc) -- what used to be four row kernels per layer around the attention:
// EP_PACK (with EP_PREAFF, N = 3 width: the fused q | k | v projection): besides the output, the copies the attention kernels stream --
//   of the Q third the rows scaled by aux.qscale (aux.qs, row-major, width columns) and their transpose (aux.qst[c][pos(m)], row length
//   aux.ldqst: all images side by side), of the K and V thirds the transposes per image (aux.kt / aux.vt [image][c][pos(l)], row length
//   aux.lt = tokens per image); pos() as under EP_TRANS.  Any of the four may be null.  M % 64 == 0, aux.lt % 64 == 0.
// EP_DELTA (with EP_TRANS; R = the attention's output O, ldr its row length): aux.ndelta[m * heads + n0 / 64] = - sum over the part's 64
//   columns of (rounded output) x O -- the delta of the attention backward, from the GEMM that produces dO.
// (Merging EP_STATS' records inside the consuming GEMM instead of k_geo_rowstat_finish was tried and dropped: at one tile per CU the merge sits
//   in the tile's exposed prologue and cost 4.3-6.5 us per launch against the 5.4 us of the kernel it replaced -- NOTEBOOK round 6.)
constexpr int EP_PACK = 1024, EP_DELTA = 2048;
struct EpiAux {
    h16 *qs = nullptr, *qst = nullptr, *kt = nullptr, *vt = nullptr;
    int lt = 0, ldqst = 0;
    float qscale = 1.0f;
    float* ndelta = nullptr;
    int heads = 0;
};

__device__ __forceinline__ float gelu_grad(float v) {   // d/dv [0.5 v (1 + erf(v / sqrt 2))] = Phi(v) + v phi(v)
    const float x = v * 0.70710678118654752f, ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float g = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);   // exp(-v^2 / 2)
    const float erfa = 1.0f - p * t * g;
    const float cdf = 0.5f + 0.5f * copysignf(erfa, v);
    return cdf + v * 0.3989422804014327f * g;
}

// sum over the 8 consecutive lanes that hold one row of the epilogue's read-back (all of them get it): three DPP adds -- quad_perm
// [1,0,3,2], [2,3,0,1], row_half_mirror -- where __shfl_xor is three ds_bpermute round trips through the LDS pipe
__device__ __forceinline__ float sum8(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true));
    return x;
}

// What an epilogue part reads from global memory, fetched AHEAD of its use: at one 256 x 256 tile per CU nothing hides an epilogue's
// load round trips (1-1.5 us each under the load of the neighbouring tiles' DMA; four of them per tile -- bias and residual of two
// parts -- were 15 % of a K = 1024 tile), so the kernels issue the column vectors and the first part's rows right after the main
// loop and the second part's rows before they work on the first.
struct EpiCols {   // per lane: bias (and, EP_PREAFF, the folded weights' row sums) of its 4 consecutive columns in each of the 8 groups;
    f32x4 b[8], s[8], g0, g1;   // EP_LOGIT: gamma_post w_out of the 8 columns it reads back
};
struct EpiRows {   // the residual (or saved pre-activation) of the 8 rows x 8 columns it reads back; EP_PREAFF: (rstd, rstd mean) of its two rows
    half8 r[8];
    float rs[2], mr[2];
};
template <int EP>
__device__ __forceinline__ void epi_cols(EpiCols& c, const float* __restrict__ bias, int ldr, int ldc2, int n0, int lane) {
    const int hi = lane >> 5;
#pragma unroll
    for (int jn = 0; jn < 2; jn++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int nl = jn * 32 + 8 * g + 4 * hi;
            c.b[jn * 4 + g] = *reinterpret_cast<const f32x4*>(bias + n0 + nl);
            if (EP & EP_PREAFF) c.s[jn * 4 + g] = *reinterpret_cast<const f32x4*>(bias + ldr + n0 + nl);
        }
    if (EP & EP_LOGIT) {
        const int gn = n0 + (lane & 7) * 8;
        c.g0 = *reinterpret_cast<const f32x4*>(bias + 64 * ldc2 + gn), c.g1 = *reinterpret_cast<const f32x4*>(bias + 64 * ldc2 + gn + 4);
    }
}
template <int EP, int RB = 2>   // RB: 32-row blocks of the part (2: 64 x 64; 1: 32 x 64, the last part of a 192-row tile of k_geo_gemm8p)
__device__ __forceinline__ void epi_rows(EpiRows& p, const h16* __restrict__ R, int ldr, int M, int m0, int n0, int lane) {
    if (EP & EP_PREAFF) {
#pragma unroll
        for (int i = 0; i < RB; i++) {
            const int gm = min(m0 + i * 32 + (lane & 31), M - 1);
            const float2 st = reinterpret_cast<const float2*>(R)[gm];
            p.rs[i] = st.x, p.mr[i] = st.y;
        }
    }
    if ((EP & (EP_RESID | EP_GELUBWD | EP_DELTA)) && !(EP & EP_QNORM)) {
#pragma unroll
        for (int q = 0; q < 4 * RB; q++) {
            const int gm = min(m0 + q * 8 + (lane >> 3), M - 1), gn = n0 + (lane & 7) * 8;   // (rows beyond M are not stored)
            p.r[q] = *reinterpret_cast<const half8*>(R + (size_t)gm * ldr + gn);
        }
    }
}

template <int EP, int RB = 2>
__device__ __forceinline__ void gemm_epilogue64(const f32x16& t00, const f32x16& t01, const f32x16& t10, const f32x16& t11, h16* img,
                                                const EpiCols& pc, const EpiRows& pr, const h16* __restrict__ R, h16* __restrict__ C, int ldc,
                                                h16* __restrict__ C2, int ldc2, int M, float scale, int m0, int n0, int lane,
                                                const float* __restrict__ bias = nullptr, int ldr = 0, const EpiAux& aux = EpiAux{}) {
    const int hi = lane >> 5, l31 = lane & 31;
    // EP_PACK: the third of q | k | v this part lies in (uniform over the wave)
    const int pk_third = (EP & EP_PACK) ? n0 / (ldr / 3) : 0;
    // EP_QKN: which third of the fused projection this part lies in (uniform over the wave) and that third's norm parameters
    const int qkn_third = (EP & EP_QKN) ? n0 / (ldr / 3) : 0;
    const float* qkn = (EP & EP_QKN) ? bias + 2 * ldr + 132 * qkn_third : nullptr;
#pragma unroll
    for (int pass = (EP & EP_SAVEZ) ? 0 : 1; pass < 2; pass++) {   // pass 0: the pre-activation (EP_SAVEZ only); pass 1: the output
#pragma unroll
        for (int jn = 0; jn < 2; jn++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int nl = jn * 32 + 8 * g + 4 * hi;  // first of this lane's 4 consecutive columns (within the 64)
                const f32x4 b4 = pc.b[jn * 4 + g];
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const f32x16& t = jn == 0 ? (i == 0 ? t00 : t01) : (i == 0 ? t10 : t11);
                    half4 o;
                    if ((EP & EP_GELU) && pass == 1) {   // (scale is 1 on the GELU path: fc1)
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            f32x2 v = {t[4 * g + q], t[4 * g + q + 1]};
                            const f32x2 b2 = {b4[q], b4[q + 1]};
                            if (EP & EP_PREAFF) {
                                const f32x2 s2 = {pc.s[jn * 4 + g][q], pc.s[jn * 4 + g][q + 1]};
                                v = v * pr.rs[i] + (s2 * -pr.mr[i] + b2);
                            } else {
                                v = v + b2;
                            }
                            v = gelu_erf2(v) * scale;
                            const half2v h = __builtin_convertvector(v, half2v);
                            o[q] = h[0], o[q + 1] = h[1];
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            float v = t[4 * g + q];
                            if (EP & EP_PREAFF) v = v * pr.rs[i] + (pc.s[jn * 4 + g][q] * -pr.mr[i] + b4[q]);
                            else v += b4[q];
                            if (!(EP & (EP_QNORM | EP_QKN))) v *= scale;
                            o[q] = (h16)v;
                        }
                    }
                    *reinterpret_cast<half4*>(img + (i * 32 + l31) * CPAD + nl) = o;
                    if ((EP & EP_TRANS) && !(EP & EP_QNORM) && pass == 1) {
                        const int gm = m0 + i * 32 + l31;
                        if (gm < M) {
                            const int pm = (gm & ~15) | (gm & 3) | ((gm & 4) << 1) | ((gm & 8) >> 1);
#pragma unroll
                            for (int q = 0; q < 4; q++) C2[(size_t)(n0 + nl + q) * ldc2 + pm] = o[q];
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image is wave-private: no barrier, only this wave's own writes
        h16* dst = (pass == 0) ? C2 : C;
        const int ldd = (pass == 0) ? ldc2 : ldc;
#pragma unroll
        for (int q = 0; q < 4 * RB; q++) {
            const int ml = q * 8 + (lane >> 3), ch = lane & 7;
            const int gm = m0 + ml, gn = n0 + ch * 8;
            half8 v = *reinterpret_cast<const half8*>(img + ml * CPAD + ch * 8);
            if (((EP & EP_QNORM) || ((EP & EP_QKN) && qkn_third < 2)) && pass == 1) {   // the row's 64 columns sit in 8 consecutive lanes
                const float* qn = (EP & EP_QKN) ? qkn : reinterpret_cast<const float*>(R);
                float x[8], sm = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    x[e] = (float)v[e];
                    sm += x[e];
                }
                const float mean = sum8(sm) * (1.0f / 64.0f);
                float sq = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    x[e] -= mean;
                    sq += x[e] * x[e];
                }
                const float rstd = rsqrtf(sum8(sq) * (1.0f / 64.0f) + qn[128]);
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = (h16)((x[e] * rstd * qn[ch * 8 + e] + qn[64 + ch * 8 + e]) * scale);
                if ((EP & EP_TRANS) && gm < M) {
                    const int pm = (gm & ~15) | (gm & 3) | ((gm & 4) << 1) | ((gm & 8) >> 1);
#pragma unroll
                    for (int e = 0; e < 8; e++) C2[(size_t)(gn + e) * ldc2 + pm] = v[e];
                }
            }
            if ((EP & EP_PACK) && pass == 1 && pk_third < 2) {   // the final values back into the image, for the transposed read below
                half8 vs = v;
                if (pk_third == 0) {   // the scaled copy of Q (rounded to fp16 once more, as k_geo_transpose_perm did)
#pragma unroll
                    for (int e = 0; e < 8; e++) vs[e] = (h16)((float)v[e] * aux.qscale);
                    if (aux.qs) *reinterpret_cast<half8*>(aux.qs + (size_t)gm * (ldr / 3) + gn) = vs;
                }
                *reinterpret_cast<half8*>(img + ml * CPAD + ch * 8) = vs;
            }
            if ((EP & EP_DELTA) && pass == 1) {
                const half8 r = pr.r[q];
                float dot = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; e++) dot = __builtin_fmaf((float)v[e], (float)r[e], dot);
                dot = sum8(dot);
                if (ch == 0 && gm < M) aux.ndelta[(size_t)gm * aux.heads + (n0 >> 6)] = -dot;
            }
            if ((EP & (EP_RESID | EP_GELUBWD)) && !(EP & EP_QNORM) && pass == 1) {
                const half8 r = pr.r[q];
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = (EP & EP_RESID) ? (h16)((float)v[e] + (float)r[e]) : (h16)((float)v[e] * gelu_grad((float)r[e]));
            }
            // (EP_QKN's pre-activation copy: V is not normalised, nothing reads its copy -- that third of C2 stays unwritten)
            if (gm < M && !(EP & EP_LOGIT) && !((EP & EP_QKN) && pass == 0 && qkn_third == 2)) *reinterpret_cast<half8*>(dst + (size_t)gm * ldd + gn) = v;
            if ((EP & (EP_STATS | EP_LOGIT)) && pass == 1) {   // the row's 64 (rounded) values sit in 8 consecutive lanes
                float x[8], sm = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    x[e] = (float)v[e];
                    sm += x[e];
                }
                const float mean = sum8(sm) * (1.0f / 64.0f);
                float sq = 0.0f, dot = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float d = x[e] - mean;
                    sq = __builtin_fmaf(d, d, sq);
                }
                sq = sum8(sq);
                if (EP & EP_LOGIT) {
#pragma unroll
                    for (int e = 0; e < 4; e++) dot = __builtin_fmaf(x[e], pc.g0[e], __builtin_fmaf(x[4 + e], pc.g1[e], dot));
                    dot = sum8(dot);
                }
                if (ch == 0 && gm < M) {
                    float* st = reinterpret_cast<float*>(C2);
                    const size_t at = (size_t)gm * ldc2 + (n0 >> 6);
                    if (EP & EP_LOGIT) *reinterpret_cast<f32x4*>(st + at * 4) = f32x4{mean, sq, dot, 0.0f};
                    else *reinterpret_cast<float2*>(st + at * 2) = float2{mean, sq};
                }
            }
        }
        if ((EP & EP_PACK) && pass == 1) {
            // the part's transpose: lane = column, 16 rows of a block in operand order -> 32 contiguous bytes of the column's row
            const int w3 = ldr / 3, cn = n0 - pk_third * w3 + lane, im = m0 / aux.lt;
            h16* T = pk_third == 0 ? aux.qst : (pk_third == 1 ? aux.kt : aux.vt);
            const size_t ldT = pk_third == 0 ? (size_t)aux.ldqst : (size_t)aux.lt;
            const size_t at = pk_third == 0 ? (size_t)m0 : (size_t)im * w3 * aux.lt + (size_t)(m0 - im * aux.lt);
            if (T && m0 < M) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int blk = 0; blk < 2 * RB; blk++) {
                    half8 lo, hi8;
#pragma unroll
                    for (int pos = 0; pos < 16; pos++) {
                        const int kq = (pos & 3) | ((pos & 4) << 1) | ((pos & 8) >> 1);
                        const h16 x = img[(blk * 16 + kq) * CPAD + lane];
                        if (pos < 8) lo[pos] = x;
                        else hi8[pos - 8] = x;
                    }
                    h16* d = T + (size_t)cn * ldT + at + blk * 16;
                    *reinterpret_cast<half8*>(d) = lo;
                    *reinterpret_cast<half8*>(d + 8) = hi8;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image is reused
    }
}

// LDS byte address of a __shared__ object (for the inline-asm reads below)
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
#define GEO_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int EP>
__global__ __launch_bounds__(256, 2) void k_geo_gemm(const h16* __restrict__ A, int lda, const h16* __restrict__ Wt, int ldw,
                                                     const float* __restrict__ bias, const h16* __restrict__ R, int ldr,
                                                     h16* __restrict__ C, int ldc, int M, int N, int K, float scale, h16* __restrict__ C2,
                                                     int ldc2, const int* __restrict__ Mdev, EpiAux aux = EpiAux{}) {
    __shared__ uint4 lds[2][2][GM * GK * 2 / 16];  // [buffer][A | W][128 rows x 8 chunks] = 64 KB
    if (Mdev) M = min(M, *Mdev);   // device-resident row count (foho_geo_decode_bwd_rows): tiles beyond it leave at once
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    // XCD-aware tile order: the blocks of one XCD (L mod 8) sweep N inside one row panel of A, eight panels (one per XCD) at a time
    const int ntn = N / GN, ntm = (M + GM - 1) / GM;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int mp = (j / ntn) * 8 + xcd, nt = j % ntn;
    if (mp >= ntm) return;
    const int m0 = mp * GM, n0 = nt * GN;
    const int wr = w >> 1, wc = w & 1;  // this wave's 64 x 64 part of the tile

    // Staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 8 rows of a tile per instruction, destination lane-linear,
    // so the XOR swizzle sits on the SOURCE address): no VGPR -> LDS store pass -- register staging had this kernel bound by
    // the LDS (ds_write_b128 costs 13 cycles per wave-instruction: 832 of them + 512 of fragment reads per K-tile and CU
    // against 1024 cycles of MFMA).  The fragment reads are INLINE ASM: hipcc cannot tell the DMA's destination buffer
    // from the one being read and waits vmcnt(0) in front of every compiler-visible ds_read, which serialises the
    // prefetch with the MFMAs (measured: the first version of this kernel); reads it cannot see get no such wait, and
    // this code counts lgkmcnt itself.  A wave moves pieces w*4 .. w*4+3 of both operands per K-tile.
    const int srow = lane >> 3, sslot = lane & 7;
    const h16* asrc[4];
    const h16* wsrc[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int row = (w * 4 + p) * 8 + srow;
        const int c = sslot ^ swz(row);
        asrc[p] = A + (size_t)min(m0 + row, M - 1) * lda + c * 8;
        wsrc[p] = Wt + (size_t)(n0 + row) * ldw + c * 8;
    }

    f32x16 acc[2][2];  // [n tile][m tile]: D rows = n, D columns = m (the lane holds 4 consecutive n for one m)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    // fragment byte addresses inside buffer 0: row r, chunk c -> r * 128 + (c ^ swz(r)) * 16; the second tile of a wave
    // (rows + 32, same swizzle) is an immediate offset of 4096, the W operand one of 16384
    const int ra = wr * 64 + l31, rw = wc * 64 + l31;
    const unsigned base = lds_addr(&lds[0][0][0]);
    unsigned aa[4], aw[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        aa[kk] = base + ra * 128 + (((2 * kk + hi) ^ swz(ra)) << 4);
        aw[kk] = base + rw * 128 + (((2 * kk + hi) ^ swz(rw)) << 4);
    }

    const int nk = K / GK;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        glds16(asrc[p], &lds[0][0][(w * 4 + p) * 64]);
        glds16(wsrc[p], &lds[0][1][(w * 4 + p) * 64]);
    }
    for (int t = 0; t < nk; t++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile t has landed for every wave, and everybody is done reading the other buffer
        if (t + 1 < nk) {
            const int nb = (t + 1) & 1, k0 = (t + 1) * GK;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                glds16(asrc[p] + k0, &lds[nb][0][(w * 4 + p) * 64]);
                glds16(wsrc[p] + k0, &lds[nb][1][(w * 4 + p) * 64]);
            }
        }
        asm volatile("" ::: "memory");
        const unsigned bo = (unsigned)(t & 1) << 15;  // 32 KB per buffer
        half8 fa[2][2], fw[2][2];  // [parity of kk][tile]: the fragments of step kk + 1 are requested before step kk's MFMAs issue
        {
            const unsigned pa = aa[0] + bo, pw = aw[0] + bo;
            GEO_DSR(fa[0][0], pa, 0);
            GEO_DSR(fa[0][1], pa, 4096);
            GEO_DSR(fw[0][0], pw, 16384);
            GEO_DSR(fw[0][1], pw, 16384 + 4096);
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            if (kk < 3) {
                const unsigned pa = aa[kk + 1] + bo, pw = aw[kk + 1] + bo;
                GEO_DSR(fa[(kk + 1) & 1][0], pa, 0);
                GEO_DSR(fa[(kk + 1) & 1][1], pa, 4096);
                GEO_DSR(fw[(kk + 1) & 1][0], pw, 16384);
                GEO_DSR(fw[(kk + 1) & 1][1], pw, 16384 + 4096);
                // LDS returns in order: at most the four reads just issued may still be out
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[kk & 1][0]), "+v"(fa[kk & 1][1]), "+v"(fw[kk & 1][0]), "+v"(fw[kk & 1][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[kk & 1][0]), "+v"(fa[kk & 1][1]), "+v"(fw[kk & 1][0]), "+v"(fw[kk & 1][1]));
            }
#pragma unroll
            for (int jn = 0; jn < 2; jn++)
#pragma unroll
                for (int i = 0; i < 2; i++)
                    acc[jn][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk & 1][jn], fa[kk & 1][i], acc[jn][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();  // every wave is done with the staging buffers: they become the epilogue's transpose image

    // ---- epilogue (gemm_epilogue64): the staging buffers become the waves' transpose images
    h16* img = reinterpret_cast<h16*>(&lds[0][0][0]) + w * (64 * CPAD);
    EpiCols pc;
    EpiRows pr;
    epi_cols<EP>(pc, bias, ldr, ldc2, n0 + wc * 64, lane);
    epi_rows<EP>(pr, R, ldr, M, m0 + wr * 64, n0 + wc * 64, lane);
    gemm_epilogue64<EP>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], img, pc, pr, R, C, ldc, C2, ldc2, M, scale, m0 + wr * 64, n0 + wc * 64, lane, bias, ldr, aux);
}

// ------------------------------------------------------------------------------------------------
// The 128 x 128 x 64 GEMM with a FOUR-deep LDS ring (round 6), for launches that put at most one workgroup on a CU -- the N = 1024
// products of the ShapeVAE transformer at M = 3072 (192 tiles on 256 CUs: c_proj, fc2 and the three transposed-weight GEMMs of the
// backward).  k_geo_gemm issues a tile's eight LDS-DMA pieces per wave in one block in front of the tile's matrix instructions; a wave
// is held ~80 cycles per piece, and with ONE wave per SIMD (one workgroup per CU) nothing else feeds the matrix pipe meanwhile:
// 0.65 us per K tile against 0.21 us of matrix work (0.93 us inside the transformer chain: 59 us for K = 4096).  A deeper ring ALONE
// changed nothing (measured: the latency was never the limit).  Here tile t + 3 is issued while tile t is multiplied (4 x 32 KB of
// LDS) and a wave waits for
// its OWN oldest tile with a counted s_waitcnt vmcnt(16) (two newer tiles of 8 pieces stay in flight) before the barrier that
// publishes the tile to the other waves.  Same fragments, same epilogue as k_geo_gemm.
// ------------------------------------------------------------------------------------------------
template <int EP>
__global__ __launch_bounds__(256, 1) void k_geo_gemm_d4(const h16* __restrict__ A, int lda, const h16* __restrict__ Wt, int ldw,
                                                        const float* __restrict__ bias, const h16* __restrict__ R, int ldr,
                                                        h16* __restrict__ C, int ldc, int M, int N, int K, float scale, h16* __restrict__ C2,
                                                        int ldc2, const int* __restrict__ Mdev, EpiAux aux = EpiAux{}) {
    __shared__ uint4 lds[4][2][GM * GK * 2 / 16];  // [stage][A | W][128 rows x 8 chunks] = 128 KB
    if (Mdev) M = min(M, *Mdev);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int ntn = N / GN, ntm = (M + GM - 1) / GM;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int mp = (j / ntn) * 8 + xcd, nt = j % ntn;
    if (mp >= ntm) return;
    const int m0 = mp * GM, n0 = nt * GN;
    const int wr = w >> 1, wc = w & 1;
    const int srow = lane >> 3, sslot = lane & 7;
    const h16* asrc[4];
    const h16* wsrc[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int row = (w * 4 + p) * 8 + srow;
        const int c = sslot ^ swz(row);
        asrc[p] = A + (size_t)min(m0 + row, M - 1) * lda + c * 8;
        wsrc[p] = Wt + (size_t)(n0 + row) * ldw + c * 8;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;
    const int ra = wr * 64 + l31, rw = wc * 64 + l31;
    const unsigned base = lds_addr(&lds[0][0][0]);
    unsigned aa[4], aw[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        aa[kk] = base + ra * 128 + (((2 * kk + hi) ^ swz(ra)) << 4);
        aw[kk] = base + rw * 128 + (((2 * kk + hi) ^ swz(rw)) << 4);
    }
    const int nk = K / GK;
#define D4_ISSUE(t_)                                                        \
    do {                                                                    \
        const int sb_ = (t_) & 3, k0_ = (t_) * GK;                          \
        _Pragma("unroll") for (int p = 0; p < 4; p++) {                     \
            glds16(asrc[p] + k0_, &lds[sb_][0][(w * 4 + p) * 64]);          \
            glds16(wsrc[p] + k0_, &lds[sb_][1][(w * 4 + p) * 64]);          \
        }                                                                   \
    } while (0)
    for (int t = 0; t < 3 && t < nk; t++) D4_ISSUE(t);
    P8_STAMP_DECL;   // (development builds -DP8_STAMPS: per-wave sums of a K tile's segments, foho_geo_stamps.h; empty otherwise)
    for (int t = 0; t < nk; t++) {
        // this wave's pieces of tile t have landed when at most the pieces of the newer tiles in flight remain outstanding
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        P8_STAMP(1);
        // RAW barrier: __syncthreads() fences with s_waitcnt vmcnt(0) -- an LDS-DMA in flight is a pending LDS write -- and would drain the
        // ring at every tile (the first build of this kernel did: no faster than two stages).  This wave's own reads of stage (t - 1) & 3
        // completed with the lgkmcnt(0) of the previous tile's last k step.
        __builtin_amdgcn_s_barrier();  // tile t has landed for every wave, and everybody is done reading stage (t - 1) & 3 -- where tile t + 3 goes
        P8_STAMP(2);
        if (t + 3 < nk) D4_ISSUE(t + 3);   // (in ONE block: two pieces behind each k step's matrix instructions measured 55.6 against 41.8 us at K = 4096)
        P8_STAMP(3);
        asm volatile("" ::: "memory");
        const unsigned bo = (unsigned)(t & 3) << 15;  // 32 KB per stage
#ifdef D4_FILL_ONLY   // (development build: the ring's fill alone -- no fragment reads, no matrix instructions; results are garbage)
        continue;
#endif
        half8 fa[2][2], fw[2][2];
        {
            const unsigned pa = aa[0] + bo, pw = aw[0] + bo;
            GEO_DSR(fa[0][0], pa, 0);
            GEO_DSR(fa[0][1], pa, 4096);
            GEO_DSR(fw[0][0], pw, 16384);
            GEO_DSR(fw[0][1], pw, 16384 + 4096);
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            if (kk < 3) {
                const unsigned pa = aa[kk + 1] + bo, pw = aw[kk + 1] + bo;
                GEO_DSR(fa[(kk + 1) & 1][0], pa, 0);
                GEO_DSR(fa[(kk + 1) & 1][1], pa, 4096);
                GEO_DSR(fw[(kk + 1) & 1][0], pw, 16384);
                GEO_DSR(fw[(kk + 1) & 1][1], pw, 16384 + 4096);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[kk & 1][0]), "+v"(fa[kk & 1][1]), "+v"(fw[kk & 1][0]), "+v"(fw[kk & 1][1]));
                if (kk == 0) P8_STAMP(4);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[kk & 1][0]), "+v"(fa[kk & 1][1]), "+v"(fw[kk & 1][0]), "+v"(fw[kk & 1][1]));
            }
#pragma unroll
            for (int jn = 0; jn < 2; jn++)
#pragma unroll
                for (int i = 0; i < 2; i++)
                    acc[jn][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk & 1][jn], fa[kk & 1][i], acc[jn][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk == 1) P8_STAMP(5);
        }
        P8_STAMP(6);
        P8_ACC();
    }
    P8_STAMP_DUMP(w, nk);
#undef D4_ISSUE
    __syncthreads();  // every wave is done with the ring: it becomes the epilogue's transpose image
    h16* img = reinterpret_cast<h16*>(&lds[0][0][0]) + w * (64 * CPAD);
    EpiCols pc;
    EpiRows pr;
    epi_cols<EP>(pc, bias, ldr, ldc2, n0 + wc * 64, lane);
    epi_rows<EP>(pr, R, ldr, M, m0 + wr * 64, n0 + wc * 64, lane);
    gemm_epilogue64<EP>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], img, pc, pr, R, C, ldc, C2, ldc2, M, scale, m0 + wr * 64, n0 + wc * 64, lane, bias, ldr, aux);
}

// ------------------------------------------------------------------------------------------------
// k_geo_gemm_d4 with the FILL and the MATRIX work on different waves (round 6): a CU fills its LDS at ~86 GB/s (382 ns per 32 KB K tile)
// and a wave that issues LDS-DMA issues nothing else, so at one wave per SIMD fill time and matrix time add (0.68 us per K tile,
// NOTEBOOK round 6).  Eight waves: waves 4-7 only fill the four-deep ring, waves 0-3 only read fragments and multiply; one raw barrier
// per K tile hands tile t over and frees stage (t - 1) & 3.  The consumers run the epilogue.
// ------------------------------------------------------------------------------------------------
template <int EP>
__global__ __launch_bounds__(512, 1) void k_geo_gemm_pc(const h16* __restrict__ A, int lda, const h16* __restrict__ Wt, int ldw,
                                                        const float* __restrict__ bias, const h16* __restrict__ R, int ldr,
                                                        h16* __restrict__ C, int ldc, int M, int N, int K, float scale, h16* __restrict__ C2,
                                                        int ldc2, const int* __restrict__ Mdev, EpiAux aux = EpiAux{}) {
    __shared__ uint4 lds[4][2][GM * GK * 2 / 16];  // [stage][A | W][128 rows x 8 chunks] = 128 KB
    if (Mdev) M = min(M, *Mdev);
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wv >= 4;   // waves 4-7 only FILL the ring (each the eight pieces wave wv - 4 would), waves 0-3 only multiply
    const int w = wv & 3;
    const int ntn = N / GN, ntm = (M + GM - 1) / GM;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int mp = (j / ntn) * 8 + xcd, nt = j % ntn;
    if (mp >= ntm) return;
    const int m0 = mp * GM, n0 = nt * GN;
    const int wr = w >> 1, wc = w & 1;
    const int srow = lane >> 3, sslot = lane & 7;
    const h16* asrc[4];
    const h16* wsrc[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int row = (w * 4 + p) * 8 + srow;
        const int c = sslot ^ swz(row);
        asrc[p] = A + (size_t)min(m0 + row, M - 1) * lda + c * 8;
        wsrc[p] = Wt + (size_t)(n0 + row) * ldw + c * 8;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;
    const int ra = wr * 64 + l31, rw = wc * 64 + l31;
    const unsigned base = lds_addr(&lds[0][0][0]);
    unsigned aa[4], aw[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        aa[kk] = base + ra * 128 + (((2 * kk + hi) ^ swz(ra)) << 4);
        aw[kk] = base + rw * 128 + (((2 * kk + hi) ^ swz(rw)) << 4);
    }
    const int nk = K / GK;
#define D4_ISSUE(t_)                                                        \
    do {                                                                    \
        const int sb_ = (t_) & 3, k0_ = (t_) * GK;                          \
        _Pragma("unroll") for (int p = 0; p < 4; p++) {                     \
            glds16(asrc[p] + k0_, &lds[sb_][0][(w * 4 + p) * 64]);          \
            glds16(wsrc[p] + k0_, &lds[sb_][1][(w * 4 + p) * 64]);          \
        }                                                                   \
    } while (0)
    if (producer) {
        for (int t = 0; t < 3 && t < nk; t++) D4_ISSUE(t);
        for (int t = 0; t < nk; t++) {
            if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // tile t has landed (every producer waited for its pieces); the consumers are done with tile t - 1
            if (t + 3 < nk) D4_ISSUE(t + 3);
        }
        __syncthreads();
        return;
    }
    for (int t = 0; t < nk; t++) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned bo = (unsigned)(t & 3) << 15;  // 32 KB per stage
        half8 fa[2][2], fw[2][2];
        {
            const unsigned pa = aa[0] + bo, pw = aw[0] + bo;
            GEO_DSR(fa[0][0], pa, 0);
            GEO_DSR(fa[0][1], pa, 4096);
            GEO_DSR(fw[0][0], pw, 16384);
            GEO_DSR(fw[0][1], pw, 16384 + 4096);
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            if (kk < 3) {
                const unsigned pa = aa[kk + 1] + bo, pw = aw[kk + 1] + bo;
                GEO_DSR(fa[(kk + 1) & 1][0], pa, 0);
                GEO_DSR(fa[(kk + 1) & 1][1], pa, 4096);
                GEO_DSR(fw[(kk + 1) & 1][0], pw, 16384);
                GEO_DSR(fw[(kk + 1) & 1][1], pw, 16384 + 4096);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[kk & 1][0]), "+v"(fa[kk & 1][1]), "+v"(fw[kk & 1][0]), "+v"(fw[kk & 1][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[kk & 1][0]), "+v"(fa[kk & 1][1]), "+v"(fw[kk & 1][0]), "+v"(fw[kk & 1][1]));
            }
#pragma unroll
            for (int jn = 0; jn < 2; jn++)
#pragma unroll
                for (int i = 0; i < 2; i++)
                    acc[jn][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk & 1][jn], fa[kk & 1][i], acc[jn][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef D4_ISSUE
    __syncthreads();  // every wave is done with the ring: it becomes the epilogue's transpose image
    h16* img = reinterpret_cast<h16*>(&lds[0][0][0]) + w * (64 * CPAD);
    EpiCols pc;
    EpiRows pr;
    epi_cols<EP>(pc, bias, ldr, ldc2, n0 + wc * 64, lane);
    epi_rows<EP>(pr, R, ldr, M, m0 + wr * 64, n0 + wc * 64, lane);
    gemm_epilogue64<EP>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], img, pc, pr, R, C, ldc, C2, ldc2, M, scale, m0 + wr * 64, n0 + wc * 64, lane, bias, ldr, aux);
}

// ------------------------------------------------------------------------------------------------
// The same GEMM on 256 x 256 x 64 tiles, 8 waves x (128 x 64): per K-tile a wave still issues 8 LDS-DMA pieces, but 64 MFMAs
// instead of 16 -- on the 128-wide tile the DMA issue (60-180 cycles per piece beside MFMAs) cost as much as the matrix work
// it fed -- and 6 fragment reads per 8 MFMAs instead of 4 per 4.  128 KB of LDS, one workgroup per CU (two waves per SIMD).
// N % 256 == 0.  The epilogue's transpose image takes the wave's 128 rows in two halves.
// ------------------------------------------------------------------------------------------------
constexpr int HM = 256, HN = 256;

template <int EP>
__global__ __launch_bounds__(512, 1) void k_geo_gemm256(const h16* __restrict__ A, int lda, const h16* __restrict__ Wt, int ldw,
                                                        const float* __restrict__ bias, const h16* __restrict__ R, int ldr,
                                                        h16* __restrict__ C, int ldc, int M, int N, int K, float scale, h16* __restrict__ C2,
                                                        int ldc2, const int* __restrict__ Mdev, EpiAux aux = EpiAux{}) {
    __shared__ uint4 lds[2][2][HM * GK * 2 / 16];  // [buffer][A | W][256 rows x 8 chunks] = 128 KB
    if (Mdev) M = min(M, *Mdev);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int ntn = N / HN, ntm = (M + HM - 1) / HM;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int mp = (j / ntn) * 8 + xcd, nt = j % ntn;
    if (mp >= ntm) return;
    const int m0 = mp * HM, n0 = nt * HN;
    const int wr = w >> 2, wc = w & 3;  // this wave's 128 (M) x 64 (N) part of the tile

    const int srow = lane >> 3, sslot = lane & 7;
    const h16* asrc[4];
    const h16* wsrc[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int row = (w * 4 + p) * 8 + srow;
        const int c = sslot ^ swz(row);
        asrc[p] = A + (size_t)min(m0 + row, M - 1) * lda + c * 8;
        wsrc[p] = Wt + (size_t)(n0 + row) * ldw + c * 8;
    }

    f32x16 acc[2][4];  // [n tile][m tile]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    const int ra = wr * 128 + l31, rw = wc * 64 + l31;
    const unsigned base = lds_addr(&lds[0][0][0]);
    unsigned aa[4], aw[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        aa[kk] = base + ra * 128 + (((2 * kk + hi) ^ swz(ra)) << 4);
        aw[kk] = base + rw * 128 + (((2 * kk + hi) ^ swz(rw)) << 4);
    }

    const int nk = K / GK;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        glds16(asrc[p], &lds[0][0][(w * 4 + p) * 64]);
        glds16(wsrc[p], &lds[0][1][(w * 4 + p) * 64]);
    }
#define GEO_RD6(set, kq)                                          \
    do {                                                          \
        const unsigned pa_ = aa[kq] + bo, pw_ = aw[kq] + bo;      \
        GEO_DSR(fa[set][0], pa_, 0);                              \
        GEO_DSR(fa[set][1], pa_, 4096);                           \
        GEO_DSR(fa[set][2], pa_, 8192);                           \
        GEO_DSR(fa[set][3], pa_, 12288);                          \
        GEO_DSR(fw[set][0], pw_, 32768);                          \
        GEO_DSR(fw[set][1], pw_, 32768 + 4096);                   \
    } while (0)
    for (int t = 0; t < nk; t++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nk) {
            const int nb = (t + 1) & 1, k0 = (t + 1) * GK;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                glds16(asrc[p] + k0, &lds[nb][0][(w * 4 + p) * 64]);
                glds16(wsrc[p] + k0, &lds[nb][1][(w * 4 + p) * 64]);
            }
        }
        asm volatile("" ::: "memory");
        const unsigned bo = (unsigned)(t & 1) << 16;  // 64 KB per buffer
        half8 fa[2][4], fw[2][2];
        GEO_RD6(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            if (kk < 3) {
                GEO_RD6((kk + 1) & 1, kk + 1);
                asm volatile("s_waitcnt lgkmcnt(6)"
                             : "+v"(fa[kk & 1][0]), "+v"(fa[kk & 1][1]), "+v"(fa[kk & 1][2]), "+v"(fa[kk & 1][3]), "+v"(fw[kk & 1][0]), "+v"(fw[kk & 1][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(fa[kk & 1][0]), "+v"(fa[kk & 1][1]), "+v"(fa[kk & 1][2]), "+v"(fa[kk & 1][3]), "+v"(fw[kk & 1][0]), "+v"(fw[kk & 1][1]));
            }
#pragma unroll
            for (int jn = 0; jn < 2; jn++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc[jn][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk & 1][jn], fa[kk & 1][i], acc[jn][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef GEO_RD6
    EpiCols pc;
    EpiRows pr[2];
    // (EP_PREAFF: two floats per row -- both parts' now: a wait for them behind the first part's stores would wait for the stores)
    if (EP & EP_PREAFF) epi_rows<EP>(pr[1], R, ldr, M, m0 + wr * 128 + 64, n0 + wc * 64, lane);
    epi_cols<EP>(pc, bias, ldr, ldc2, n0 + wc * 64, lane);
    epi_rows<EP>(pr[0], R, ldr, M, m0 + wr * 128, n0 + wc * 64, lane);
    __syncthreads();

    h16* img = reinterpret_cast<h16*>(&lds[0][0][0]) + w * (64 * CPAD);
    if (!(EP & EP_PREAFF)) epi_rows<EP>(pr[1], R, ldr, M, m0 + wr * 128 + 64, n0 + wc * 64, lane);
#pragma unroll
    for (int half = 0; half < 2; half++)
        gemm_epilogue64<EP>(acc[0][2 * half], acc[0][2 * half + 1], acc[1][2 * half], acc[1][2 * half + 1], img, pc, pr[half], R, C, ldc, C2, ldc2, M,
                            scale, m0 + wr * 128 + half * 64, n0 + wc * 64, lane, bias, ldr, aux);
}

// ------------------------------------------------------------------------------------------------
// The 256 x 256 x 64 GEMM as a PHASED loop (round 5).  Same tile, same LDS image, same fragments and epilogue as k_geo_gemm256 --
// another schedule.  k_geo_gemm256 has the two waves of every SIMD in lockstep: both issue their eight LDS-DMA pieces of a K tile at
// once (650-1500 cycles of issue during which the SIMD's matrix pipe idles, NOTEBOOK round 4).  Here the two wave groups of the
// workgroup (waves 0-3 and 4-7: one wave of each SIMD per group) run ONE BARRIER APART and a K tile is four phases of
// {load segment | barrier | compute segment | barrier}: while a SIMD's one wave multiplies (8 matrix instructions = one quadrant
// of its 128 x 64 tile, s_setprio 1) its other wave issues its fragment reads and TWO LDS-DMA pieces -- a load segment is as long
// as the partner's compute segment instead of three times as long.  The next K tile arrives in quarters, in the order in which it
// will be read (W columns 0-31 of every wave, A rows 0-63, A rows 64-127, W columns 32-63), each quarter issued three phases before
// its first read; a wave waits for its own pieces with a COUNTED s_waitcnt vmcnt(2) at the end of every compute segment (everything
// but the phase's own two pieces has landed -- the queue is never drained), the barriers publish them to the other waves.
// Staging by buffer_load_dwordx4 ... lds: resource + scalar offset per piece (the addresses of all pieces of a wave differ by
// scalars), rows beyond M read zeros (the resource's bound) instead of a clamped row.  The schedule is the "8-phase" form of
// cdna_hip_programming.md section 5 (T3 + T4 + T5) laid over this kernel's 32 x 32 x 16 fragments.
//
// 192-row tiles (template parameter TM = 192, late round 6).  A launch that is ONE round of 256-row tiles with CUs left over -- the ShapeVAE
// transformer's M = 3072 x N = 4096: 12 x 16 tiles on 256 CUs -- is bound by what one CU multiplies; 16 x 16 tiles of 192 rows give every CU
// three quarters of that (34.4 -> 28.0 us; gemm() chooses).  Same LDS layout (the A region keeps its 256 rows, 64 unused), same phases; a
// wave's part is 96 x 64: "A rows sub 0" stays two 32-row tiles, "sub 1" is ONE -- phases 1 and 2 run four matrix instructions
// (P8_COMPUTE1), the DMA quarter "A rows sub 1" is one piece per wave (piece w: group w / 4, rows 64 + 8 (w % 4); swizzle by the parity of
// w), so any four consecutive phases issue 7 pieces instead of 8 and the counted waits are vmcnt(7) (7 / 5 / 3 / 2 in the tile before last,
// vmcnt(5) behind the prologue); the wave's second epilogue part is 32 x 64 (gemm_epilogue64<EP, 1>).
// ------------------------------------------------------------------------------------------------
// (development builds -DP8_STAMPS / -DP8_TIMELINE: the P8_STAMP / P8_TL hooks below are defined in foho_geo_stamps.h; empty otherwise)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_base, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <int EP, int TM = HM>   // TM: rows of a tile, 256 or 192 (the note on 192-row tiles above)
__global__ __launch_bounds__(512, 1) void k_geo_gemm8p(const h16* __restrict__ A, int lda, const h16* __restrict__ Wt, int ldw,
                                                       const float* __restrict__ bias, const h16* __restrict__ R, int ldr,
                                                       h16* __restrict__ C, int ldc, int M, int N, int K, float scale, h16* __restrict__ C2,
                                                       int ldc2, const int* __restrict__ Mdev, EpiAux aux = EpiAux{}) {
    // [buffer][A | W][256 rows x 8 chunks] = 128 KB, + 9.5 KB: the epilogue's image (8 waves x 64 rows x 72 halfs = 72 KB) lies over buffer 1
    // and this tail, so that buffer 0 can take the NEXT tile's first K tile while the epilogue runs
    __shared__ uint4 ldsx[2 * 2 * (HM * GK * 2 / 16) + 608];
    uint4 (&lds)[2][2][HM * GK * 2 / 16] = *reinterpret_cast<uint4 (*)[2][2][HM * GK * 2 / 16]>(ldsx);
    // the epilogue's column vectors (bias; EP_PREAFF: the folded weights' row sums and the rows' statistics; EP_LOGIT: gamma w_out) wait
    // in LDS from the start of the tile: the epilogue of a K = 1024 tile has no global round trip of its own to sit out
    __shared__ float ext_b[HN], ext_s[HN];
    __shared__ float2 ext_r[HM];
    if (Mdev) M = min(M, *Mdev);
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int GR = TM / 2, S1T = (GR - 64) / 32;   // rows per wave group; 32-row tiles in its second half ("sub 1": rows 64 .. GR - 1)
    static_assert(TM == 256 || TM == 192, "k_geo_gemm8p: 256- or 192-row tiles");
    const int ntn = N / HN, ntm = (M + TM - 1) / TM;
    // PERSISTENT: a workgroup walks the tiles L = blockIdx.x, + gridDim.x, ... (the host launches one workgroup per CU when there are
    // more tiles than CUs).  Measured per tile of the fc1 shape before (shader cycles, scripts/dev_p8_timeline.py): K loop 36 900, but
    // 8 300 from entry to the first matrix instruction (the first K tile's DMA: 2 700 until this wave's pieces land + 3 900 until the
    // slowest wave's), 4 300 epilogue and 3 300 between one workgroup's exit and the next one's entry on the CU.  Here the next
    // tile's first K tile is fetched into buffer 0 WHILE the epilogue runs, and there is no exit / entry in between.
    const int total = 8 * ((ntm + 7) / 8) * ntn;
    auto tile_of = [&](int L_, int& m0_, int& n0_) {
        const int xcd = L_ & 7, j = L_ >> 3, mp = (j / ntn) * 8 + xcd;
        m0_ = mp * TM, n0_ = (j % ntn) * HN;
        return mp < ntm;
    };
    int L = blockIdx.x, m0 = 0, n0 = 0;
    while (L < total && !tile_of(L, m0, n0)) L += gridDim.x;
    if (L >= total) return;
    const int wr = w >> 2, wc = w & 3;  // this wave's 128 (M) x 64 (N) part of the tile; wr = its group
    P8_TL_DECL;
    P8_TL(0);
    float ev0 = 0.0f, ev1 = 0.0f;
    float2 ev2 = float2{1.0f, 0.0f};
#define P8_EV_LOAD(E0, E1, E2, M0, N0)                                                     \
    do {                                                                                   \
        if (tid < HN) {                                                                    \
            E0 = bias[(N0) + tid];                                                         \
            if (EP & EP_PREAFF) E1 = bias[ldr + (N0) + tid];                               \
            if (EP & EP_LOGIT) E1 = bias[64 * ldc2 + (N0) + tid];                          \
        } else if (EP & EP_PREAFF) {                                                       \
            E2 = reinterpret_cast<const float2*>(R)[min((M0) + tid - HN, M - 1)];          \
        }                                                                                  \
    } while (0)
    P8_EV_LOAD(ev0, ev1, ev2, m0, n0);
#define P8_EV_WRITE()                                                                      \
    do {                                                                                   \
        if (tid < HN) {                                                                    \
            ext_b[tid] = ev0;                                                              \
            if (EP & (EP_PREAFF | EP_LOGIT)) ext_s[tid] = ev1;                             \
        } else if (EP & EP_PREAFF) {                                                       \
            ext_r[tid - HN] = ev2;                                                         \
        }                                                                                  \
    } while (0)
    P8_EV_WRITE();   // (the first tile's vectors: one exposed round trip per workgroup; the later tiles' travel during the K loop)

    // ---- LDS-DMA pieces of this wave: per K tile two pieces in each of four phases.  A piece = 8 rows x 128 bytes; a lane's 16 bytes
    // land at (row0 + lane / 8, slot lane % 8), so it FETCHES chunk slot ^ swz(row); swz(row0 + r) depends on row0 only through
    // the parity of row0 / 8, which is the parity of the piece's index in its quarter: pieces 2w (even) and 2w + 1 (odd).
    const int srow = lane >> 3, sslot = lane & 7;
    const int va0 = (srow * lda + ((sslot ^ ((srow >> 1) & 7)) << 3)) * 2, va1 = (srow * lda + ((sslot ^ ((4 + (srow >> 1)) & 7)) << 3)) * 2;
    const int va_s1 = (w & 1) ? va1 : va0;   // the swizzle of a wave's single sub-1 piece (192-row tiles) goes by the parity of its index: w
    const int vw0 = (srow * ldw + ((sslot ^ ((srow >> 1) & 7)) << 3)) * 2, vw1 = (srow * ldw + ((sslot ^ ((4 + (srow >> 1)) & 7)) << 3)) * 2;
    // tile rows of the pieces, by quarter: W columns sub 0 (phase 0), A rows sub 0 (phase 1), A rows sub 1 (phase 2), W columns sub 1 (phase 3)
    int rowq[4][2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int Lp = 2 * w + e;
        rowq[0][e] = 64 * (Lp >> 2) + 8 * (Lp & 3);
        rowq[1][e] = GR * (Lp >> 3) + 8 * (Lp & 7);
        // (192-row tiles: sub 1 is 32 rows per group = 8 pieces, ONE per wave -- piece w: group w / 4, rows 64 + 8 (w % 4); only e = 0 is used)
        rowq[2][e] = S1T == 2 ? GR * (Lp >> 3) + 64 + 8 * (Lp & 7) : GR * (w >> 2) + 64 + 8 * (w & 3);
        rowq[3][e] = 64 * (Lp >> 2) + 32 + 8 * (Lp & 3);
    }
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)A, (short)0, (int)min((size_t)M * lda * 2, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, (short)0, (int)min((size_t)N * ldw * 2, (size_t)0x7fffffff), 0x00020000);
    int sq[4][2];   // scalar byte offsets of the pieces at K tile 0
#define P8_SET_SQ(M0, N0)                                                                  \
    _Pragma("unroll") for (int e = 0; e < 2; e++) {                                        \
        sq[0][e] = ((N0) + rowq[0][e]) * ldw * 2;                                          \
        sq[1][e] = ((M0) + rowq[1][e]) * lda * 2;                                          \
        sq[2][e] = ((M0) + rowq[2][e]) * lda * 2;                                          \
        sq[3][e] = ((N0) + rowq[3][e]) * ldw * 2;                                          \
    }
    P8_SET_SQ(m0, n0);
#define P8_DMA(q, nb, koff)                                                                                                        \
    do {                                                                                                                           \
        if ((q) == 0 || (q) == 3) {                                                                                                \
            dma16(rw_, &lds[nb][1][rowq[q][0] * 8], vw0, sq[q][0] + (koff));                                                       \
            dma16(rw_, &lds[nb][1][rowq[q][1] * 8], vw1, sq[q][1] + (koff));                                                       \
        } else if ((q) == 2 && S1T == 1) {                                                                                         \
            dma16(ra_, &lds[nb][0][rowq[2][0] * 8], va_s1, sq[2][0] + (koff));                                                     \
        } else {                                                                                                                   \
            dma16(ra_, &lds[nb][0][rowq[q][0] * 8], va0, sq[q][0] + (koff));                                                       \
            dma16(ra_, &lds[nb][0][rowq[q][1] * 8], va1, sq[q][1] + (koff));                                                       \
        }                                                                                                                          \
    } while (0)

    f32x16 acc[2][4];  // [n tile][m tile]

    const int ra = wr * GR + l31, rw = wc * 64 + l31;
    const unsigned base = lds_addr(&lds[0][0][0]);
    unsigned aa[4], aw[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        aa[kk] = base + ra * 128 + (((2 * kk + hi) ^ swz(ra)) << 4);
        aw[kk] = base + rw * 128 + (((2 * kk + hi) ^ swz(rw)) << 4);
    }
    const int nk = K / GK;
    half8 fa0[4][2], fa1[4][2], fw0[4], fw1[4];   // A rows sub 0 / sub 1 (two 32-row tiles each), W columns sub 0 / sub 1, by K step

    // ---- prologue: K tile 0 whole (for every tile but the workgroup's first it is already on its way: issued before the previous
    // tile's epilogue) and the three quarters of K tile 1 the steady state would have issued by now; then W columns sub 0 of tile 0
    // into registers; the second group starts one barrier late
#pragma unroll
    for (int q = 0; q < 4; q++) P8_DMA(q, 0, 0);
  for (;;) {   // ---- one output tile per pass
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;
    // the next tile of this workgroup: its vectors travel (into registers) during this tile's K loop
    int Ln = L + gridDim.x, m0n = 0, n0n = 0;
    while (Ln < total && !tile_of(Ln, m0n, n0n)) Ln += gridDim.x;
    const bool more = Ln < total;
    if (more) P8_EV_LOAD(ev0, ev1, ev2, m0n, n0n);
    P8_DMA(0, 1, GK * 2);
    P8_DMA(1, 1, GK * 2);
    P8_DMA(2, 1, GK * 2);
    P8_TL(5);
    // (everything older than the six -- 192-row tiles: five -- pieces just issued: the first K tile, and the previous tile's stores)
    if (S1T == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    P8_TL(6);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kk = 0; kk < 4; kk++) GEO_DSR(fw0[kk], aw[kk], 32768);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fw0[0]), "+v"(fw0[1]), "+v"(fw0[2]), "+v"(fw0[3]));
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();
    P8_TL(1);

#define P8_DSR(dst, addr, off) GEO_DSR(dst, addr, off)
#define P8_LGKM "s_waitcnt lgkmcnt(0)"
#define P8_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
    // compute segment: prio 1, eight matrix instructions, prio 0, the counted wait for this wave's older DMA pieces, barrier
#define P8_COMPUTE(ACC0, ACC1, FW, FA, WAITN)                                                                                       \
    do {                                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
        P8_STAMP(3);                                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 4; kk++) {                                                                          \
            ACC0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FW[kk], FA[kk][0], ACC0, 0, 0, 0);                                        \
            ACC1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FW[kk], FA[kk][1], ACC1, 0, 0, 0);                                        \
            if (kk == 0) __builtin_amdgcn_sched_barrier(0);   /* (pins the segment: without it the scheduler moves matrix instructions across the phase's barriers) */ \
        }                                                                                                                           \
        __builtin_amdgcn_s_setprio(0);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
        P8_STAMP(4);                                                                                                                \
        P8_WAIT(WAITN);                                                                                                             \
        P8_STAMP(5);                                                                                                                \
        __builtin_amdgcn_s_barrier();                                                                                               \
        P8_STAMP(6);                                                                                                                \
        P8_ACC();                                                                                                                   \
    } while (0)
    /* ... the same with ONE 32-row tile of A (sub 1 of a 192-row tile): four matrix instructions */                                  \
#define P8_COMPUTE1(ACC0, FW, FA, WAITN)                                                                                           \
    do {                                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                                              \
        _Pragma("unroll") for (int kk = 0; kk < 4; kk++) {                                                                          \
            ACC0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FW[kk], FA[kk][0], ACC0, 0, 0, 0);                                        \
            if (kk == 0) __builtin_amdgcn_sched_barrier(0);                                                                         \
        }                                                                                                                           \
        __builtin_amdgcn_s_setprio(0);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
        P8_WAIT(WAITN);                                                                                                             \
        __builtin_amdgcn_s_barrier();                                                                                               \
    } while (0)
#define P8_SYNC8(F)                                                                                                                 \
    do {                                                                                                                            \
        P8_STAMP(1);                                                                                                                \
        __builtin_amdgcn_s_barrier();                                                                                               \
        P8_STAMP(2);                                                                                                                \
        asm volatile(P8_LGKM : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[2][0]), "+v"(F[2][1]), "+v"(F[3][0]), "+v"(F[3][1])); \
    } while (0)
#define P8_SYNC4A(F)                                                                               \
    do {                                                                                           \
        __builtin_amdgcn_s_barrier();                                                              \
        asm volatile(P8_LGKM : "+v"(F[0][0]), "+v"(F[1][0]), "+v"(F[2][0]), "+v"(F[3][0]));        \
    } while (0)
#define P8_SYNC4(F)                                                                                \
    do {                                                                                           \
        P8_STAMP(1);                                                                               \
        __builtin_amdgcn_s_barrier();                                                              \
        P8_STAMP(2);                                                                               \
        asm volatile(P8_LGKM : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]));                    \
    } while (0)
    // One K tile = four phases.  Which DMA quarter a phase issues: the one whose LDS region the workgroup has finished reading two
    // phases earlier (both groups: the second group's reads of a region retire one barrier after the first's), SIX phases ahead of
    // its own first read -- phase 0: W columns sub 1 of tile t + 1; phase 1: W columns sub 0 of t + 2; phase 2: A rows sub 0 of
    // t + 2; phase 3: A rows sub 1 of t + 2 (tile t + 2 goes into THIS tile's buffer).  I0..I3: does the phase issue (the last two
    // tiles issue less); W0..W3: pieces of this wave that may still be in flight at the end of the phase's compute segment =
    // 2 x (issuing phases among the last four): everything older has landed, and two barriers later every wave knows it.
#define P8_TILE(I0, I1, I2, I3, W0, W1, W2, W3)                                                                                     \
    do {                                                                                                                            \
        const unsigned bo = (unsigned)(t & 1) << 16, bn = bo ^ 0x10000u;   /* 64 KB per buffer */                                   \
        const int cb = t & 1, nb = cb ^ 1, k1 = (t + 1) * GK * 2, k2 = (t + 2) * GK * 2;                                            \
        /* phase 0: A rows sub 0 (x) W columns sub 0 (read in the previous phase 3) */                                              \
        _Pragma("unroll") for (int kk = 0; kk < 4; kk++) {                                                                          \
            P8_DSR(fa0[kk][0], aa[kk] + bo, 0);                                                                                     \
            P8_DSR(fa0[kk][1], aa[kk] + bo, 4096);                                                                                  \
        }                                                                                                                           \
        if (I0) P8_DMA(3, nb, k1);                                                                                                  \
        P8_SYNC8(fa0);                                                                                                              \
        P8_COMPUTE(acc[0][0], acc[0][1], fw0, fa0, W0);                                                            \
        /* phase 1: A rows sub 1 (x) W columns sub 0 (sub 1 of a 192-row tile is ONE 32-row tile: four matrix instructions) */      \
        _Pragma("unroll") for (int kk = 0; kk < 4; kk++) {                                                                          \
            P8_DSR(fa1[kk][0], aa[kk] + bo, 8192);                                                                                  \
            if (S1T == 2) P8_DSR(fa1[kk][1], aa[kk] + bo, 12288);                                                                   \
        }                                                                                                                           \
        if (I1) P8_DMA(0, cb, k2);                                                                                                  \
        if (S1T == 2) {                                                                                                             \
            P8_SYNC8(fa1);                                                                                                          \
            P8_COMPUTE(acc[0][2], acc[0][3], fw0, fa1, W1);                                                                         \
        } else {                                                                                                                    \
            P8_SYNC4A(fa1);                                                                                                         \
            P8_COMPUTE1(acc[0][2], fw0, fa1, W1);                                                                                   \
        }                                                                                                                           \
        /* phase 2: A rows sub 1 (x) W columns sub 1 */                                                                             \
        _Pragma("unroll") for (int kk = 0; kk < 4; kk++) P8_DSR(fw1[kk], aw[kk] + bo, 32768 + 4096);                                \
        if (I2) P8_DMA(1, cb, k2);                                                                                                  \
        P8_SYNC4(fw1);                                                                                                              \
        if (S1T == 2) P8_COMPUTE(acc[1][2], acc[1][3], fw1, fa1, W2);                                                               \
        else P8_COMPUTE1(acc[1][2], fw1, fa1, W2);                                                                                  \
        /* phase 3: A rows sub 0 (x) W columns sub 1; W columns sub 0 of the NEXT tile come in for its phase 0 */                   \
        if (I0) {                                                                                                                   \
            _Pragma("unroll") for (int kk = 0; kk < 4; kk++) P8_DSR(fw0[kk], aw[kk] + bn, 32768);                                   \
        }                                                                                                                           \
        if (I3) P8_DMA(2, cb, k2);                                                                                                  \
        P8_SYNC4(fw0);                                                                                                              \
        P8_COMPUTE(acc[1][0], acc[1][1], fw1, fa0, W3);                                                            \
    } while (0)

    P8_STAMP_DECL;
    int t = 0;
    if (S1T == 2) {
        for (; t < nk - 2; t++) P8_TILE(1, 1, 1, 1, 8, 8, 8, 8);
        P8_TILE(1, 0, 0, 0, 8, 6, 4, 2);   // t = nk - 2: only W columns sub 1 of the last tile is still to come
    } else {   // 192-row tiles: the phase that issues A rows sub 1 issues ONE piece -- 7 in any four phases
        for (; t < nk - 2; t++) P8_TILE(1, 1, 1, 1, 7, 7, 7, 7);
        P8_TILE(1, 0, 0, 0, 7, 5, 3, 2);
    }
    t++;
    P8_TILE(0, 0, 0, 0, 0, 0, 0, 0);   // t = nk - 1
    P8_STAMP_DUMP(w, nk);
    P8_TL(2);
    // ---- the epilogue's column vectors and row statistics from LDS into registers ...
    EpiCols pc;
    EpiRows pr[2];
#pragma unroll
    for (int g8 = 0; g8 < 8; g8++) {
        const int nl = wc * 64 + (g8 >> 2) * 32 + 8 * (g8 & 3) + 4 * hi;
        pc.b[g8] = *reinterpret_cast<const f32x4*>(&ext_b[nl]);
        if (EP & EP_PREAFF) pc.s[g8] = *reinterpret_cast<const f32x4*>(&ext_s[nl]);
    }
    if (EP & EP_LOGIT) {
        pc.g0 = *reinterpret_cast<const f32x4*>(&ext_s[wc * 64 + (lane & 7) * 8]);
        pc.g1 = *reinterpret_cast<const f32x4*>(&ext_s[wc * 64 + (lane & 7) * 8 + 4]);
    }
    if (EP & EP_PREAFF) {
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const float2 st = ext_r[wr * GR + half * 64 + i * 32 + l31];   // (192-row tiles: half 1, i = 1 is read and not used)
                pr[half].rs[i] = st.x, pr[half].mr[i] = st.y;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (wr == 0) __builtin_amdgcn_s_barrier();   // the first group's count catches up with the second's
    __syncthreads();                                              // every wave is done with both buffers and has the vectors in registers
    P8_TL(3);
    // ... the next tile's into their place (nothing of this wave's is in flight here: the wait the compiler puts in front is free) ...
    if (more) P8_EV_WRITE();
    // ... the residual rows of BOTH parts from global memory (a later load would queue behind the next tile's DMA: vmcnt is in order) ...
    if (!(EP & EP_PREAFF)) {
        epi_rows<EP>(pr[0], R, ldr, M, m0 + wr * GR, n0 + wc * 64, lane);
        epi_rows<EP, S1T>(pr[1], R, ldr, M, m0 + wr * GR + 64, n0 + wc * 64, lane);
    }
    const int m0c = m0, n0c = n0;
    if (more) {   // ... and the next tile's first K tile into buffer 0, on its way during the epilogue
        m0 = m0n, n0 = n0n;
        P8_SET_SQ(m0, n0);
#pragma unroll
        for (int q = 0; q < 4; q++) P8_DMA(q, 0, 0);
    }

    h16* img = reinterpret_cast<h16*>(&lds[1][0][0]) + w * (64 * CPAD);
    gemm_epilogue64<EP>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], img, pc, pr[0], R, C, ldc, C2, ldc2, M, scale, m0c + wr * GR, n0c + wc * 64, lane, bias, ldr, aux);
    // (the wave's second part: 64 rows of a 256-row tile, 32 of a 192-row tile)
    gemm_epilogue64<EP, S1T>(acc[0][2], acc[0][3], acc[1][2], acc[1][3], img, pc, pr[1], R, C, ldc, C2, ldc2, M, scale, m0c + wr * GR + 64, n0c + wc * 64, lane, bias,
                             ldr, aux);
    P8_TL(4);
    P8_TL_DUMP(L);
    if (!more) break;
    __syncthreads();   // the image (buffer 1) and the vectors' LDS slots are free for the next tile
    L = Ln;
    P8_TL(0);
  }
#undef P8_TILE
#undef P8_COMPUTE
#undef P8_SYNC8
#undef P8_SYNC4
#undef P8_SYNC4A
#undef P8_COMPUTE1
#undef P8_DMA
#undef P8_WAIT
#undef P8_DSR
#undef P8_EV_LOAD
#undef P8_EV_WRITE
#undef P8_SET_SQ
}

// ------------------------------------------------------------------------------------------------
// Cross attention, head dimension 64: O[M, heads*64] = softmax(Q K^T) V per head, Q pre-scaled by log2(e)/sqrt(64).
// K rows: Kp + l * ldk + head * 64 (the K half of the KV projection, as the GEMM left it); V: Vt[head][d][pos(l)],
// transposed, with the keys of every 16-block stored in the order the P fragment holds them (pack_vt below).
// Workgroup = 4 waves x 64 queries of ONE head; 64-key tiles of K and V^T double-buffered in LDS, brought by LDS-DMA (buffer_load ... lds:
// tile t + 1 is in flight while tile t is computed; no staging registers, which is what pays for four score buffers -- see inside).
// ------------------------------------------------------------------------------------------------
// Images of a batch in ONE launch (round 6: the ShapeVAE transformer's attention for B images; a launch per image leaves the chip
// under-filled -- 192 / 384 workgroups at 16 heads x 3072 tokens).  Per operand of a kernel, the distance between two images in
// elements of the operand's type; the image is blockIdx.y (k_geo_pack_vt: blockIdx.z).  All zero with gridDim.y == 1: one image.
struct AttnB {
    long a = 0, b = 0, c = 0, d = 0, e = 0, f = 0, g = 0, h = 0;
};
constexpr int AQ = 256, AK = 64;
constexpr float RESCALE_THR = 6.0f;  // log2 domain: P <= 2^6 while the running max lags behind

// value of the same register in the other half-wave (lane ^ 32): v_permlane32_swap exchanges vdst[32..63] with src[0..31]
__device__ __forceinline__ float other_half(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
// three-input maximum: v_max3_f32.  Plain fmaxf() -- NOT inline asm: hipcc pads the MFMA -> VALU read hazard only for
// instructions it can see, and an asm v_max3 on fresh accumulators read them before the matrix pipe had written them
// (results stayed accurate but differed from run to run).  The translation unit is built with -fno-honor-nans, which is
// what keeps the compiler from putting a canonicalising v_max_f32 x, x in front of every operand.
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }  // bare v_exp_f32: arguments are <= 6, underflow to 0 is what is wanted

__global__ __launch_bounds__(256, 2) void k_geo_attn(const h16* __restrict__ Q, int ldq, const h16* __restrict__ Kp, int ldk,
                                                     const h16* __restrict__ Vt, int L, h16* __restrict__ O, int ldo, int M,
                                                     int heads, float* __restrict__ nlse, const int* __restrict__ Mdev, int qhs = 64, int khs = 64,
                                                     float qscale = 1.0f, float* __restrict__ lse_nat = nullptr, AttnB bs = AttnB{}) {
    // bs: a = Q, b = Kp, c = Vt, d = O, e = nlse, f = lse_nat
    Q += blockIdx.y * bs.a, Kp += blockIdx.y * bs.b, Vt += blockIdx.y * bs.c, O += blockIdx.y * bs.d;
    if (nlse) nlse += blockIdx.y * bs.e;
    if (lse_nat) lse_nat += blockIdx.y * bs.f;
    // lse_nat: optional (heads, M) fp32 -- the NATURAL-log log-sum-exp of the scaled scores per head and query, the form torch's
    // attention backward kernels take (foho_sdpa_fwd)
    // qhs / khs: distance of two heads inside a row of Q / K (64: heads side by side; 192: hy3dgen's interleaved q | k | v per head);
    // qscale != 1: Q arrives UNSCALED and is multiplied (and rounded to fp16 again) as it is loaded (foho_sdpa_fwd)
    __shared__ uint4 lds[2][2][AK * 8];  // [buffer][K | Vt][64 rows x 8 chunks] = 32 KB
    if (Mdev) M = min(M, *Mdev);
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int head, qblk;
    if ((heads & 7) == 0) {  // an XCD (block id mod 8) keeps heads/8 heads: their K and V stay in its L2
        const int hpx = heads >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        head = xcd * hpx + j % hpx;
        qblk = j / hpx;
    } else {
        head = blockIdx.x % heads;
        qblk = blockIdx.x / heads;
    }
    if (qblk * AQ >= M) return;   // (only with a device-resident row count: the grid is sized for the capacity)
    const int q0 = qblk * AQ + w * 64;

    half8 qf[2][4];  // B operand of S^T = K Q^T: lane (q, hi) holds Q[q][16 kk + 8 hi .. + 7]
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
        const int row = min(q0 + qb * 32 + l31, M - 1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            qf[qb][kk] = *reinterpret_cast<const half8*>(Q + (size_t)row * ldq + head * qhs + 16 * kk + 8 * hi);
            if (qscale != 1.0f)
#pragma unroll
                for (int e = 0; e < 8; e++) qf[qb][kk][e] = (h16)((float)qf[qb][kk][e] * qscale);
        }
    }

    // ---- K / V^T tiles by LDS-DMA: per key tile every wave brings pieces 2 w and 2 w + 1 (8 rows x 128 bytes each) of both; a lane's
    // 16 bytes land at (row, slot = lane & 7), so it FETCHES chunk slot ^ swz(row)
    const int srow = lane >> 3, sslot = lane & 7;
    const int c0 = (sslot ^ ((srow >> 1) & 7)) << 3, c1 = (sslot ^ ((4 + (srow >> 1)) & 7)) << 3;
    const int vk0 = (srow * ldk + c0) * 2, vk1 = (srow * ldk + c1) * 2, vt0 = (srow * L + c0) * 2, vt1 = (srow * L + c1) * 2;
    const size_t kspan = ((size_t)(L - 1) * ldk + (size_t)(heads - 1) * khs + 64) * 2;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, (short)0, (int)min(kspan, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vt, (short)0, (int)min((size_t)heads * 64 * L * 2, (size_t)0x7fffffff), 0x00020000);
    const int r0 = 16 * w, r1 = 16 * w + 8;
    const int sk = head * khs * 2, skt = head * 64 * L * 2;
#define A2_ISSUE(t, buf)                                                              \
    do {                                                                              \
        const int kb_ = (t) * AK;                                                     \
        dma16(rk, &lds[buf][0][r0 * 8], vk0, sk + (kb_ + r0) * ldk * 2);              \
        dma16(rk, &lds[buf][0][r1 * 8], vk1, sk + (kb_ + r1) * ldk * 2);              \
        dma16(rv, &lds[buf][1][r0 * 8], vt0, skt + (r0 * L + kb_) * 2);               \
        dma16(rv, &lds[buf][1][r1 * 8], vt1, skt + (r1 * L + kb_) * 2);               \
    } while (0)
    const unsigned lbase = lds_addr(&lds[0][0][0]);
    unsigned ak[4], av[2][2];   // fragment addresses in buffer 0: K rows (row = key; + 4096 per sub-tile), V^T rows (row = d; + 4096 per d tile)
#pragma unroll
    for (int kk = 0; kk < 4; kk++) ak[kk] = lbase + l31 * 128 + (((2 * kk + hi) ^ swz(l31)) << 4);
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) av[sub][k2] = lbase + 8192 + l31 * 128 + (((4 * sub + 2 * k2 + hi) ^ swz(l31)) << 4);

    f32x16 o[2][2];  // [query block][d tile]: O^T, rows d, columns q
    f32x16 negm[2];  // -(running max) of the lane's query in all 16 registers: the accumulator S^T starts from
#pragma unroll
    for (int a = 0; a < 2; a++) {
#pragma unroll
        for (int r = 0; r < 16; r++) negm[a][r] = 0.0f;
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[a][b][r] = 0.0f;
    }
    float lsum[2] = {0.0f, 0.0f};

    // One key tile = four UNITS (32 keys x 32 queries): u0 = (sub-tile 0, query block 0), u1 = (0, 1), u2 = (1, 0), u3 = (1, 1).  A unit is QK
    // (4 matrix instructions) -> softmax -> PV (4).  FOUR score buffers and two P buffers: while unit u is exponentiated the matrix pipe
    // has the PV of unit u - 1 and the QK of unit u + 2 -- both independent of u -- instead of only work that waits for u's own P.
    //   R0: QK u0, QK u1        R1: check u0; exp u0 || QK u2        R2: check u1; exp u1 || PV u0, QK u3
    //   R3: check u2; exp u2 || PV u1    R4: check u3; exp u3 || PV u2    R5: PV u3
    // (the QK of a query block starts from -(its running max): it is issued after the check of that block's previous unit, as before)
    auto qk = [&](const half8* kf, int qb) {
        f32x16 r = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[qb][0], negm[qb], 0, 0, 0);
#pragma unroll
        for (int kk = 1; kk < 4; kk++) r = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kk], qf[qb][kk], r, 0, 0, 0);
        return r;
    };
    auto check = [&](f32x16& sc, int qb, bool first) {   // raise the running max when a score exceeds it by 2^6 (the first unit SETS it)
        float tm = max3(sc[0], sc[1], sc[2]);
        tm = max3(tm, sc[3], sc[4]);
        tm = max3(tm, sc[5], sc[6]);
        tm = max3(tm, sc[7], sc[8]);
        tm = max3(tm, sc[9], sc[10]);
        tm = max3(tm, sc[11], sc[12]);
        tm = max3(tm, sc[13], sc[14]);
        tm = fmaxf(tm, sc[15]);
        if (first || __any(tm > RESCALE_THR)) {
            tm = fmaxf(tm, other_half(tm));
            const float up = first ? tm : fmaxf(tm, 0.0f);
            const float alpha = ex2(-up);
            lsum[qb] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                sc[r] -= up;
                negm[qb][r] -= up;
            }
#pragma unroll
            for (int dt = 0; dt < 2; dt++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[qb][dt][r] *= alpha;
        }
    };
    auto expo = [&](const f32x16& sc, int qb, half8* pf) {
        float ls = 0.0f;
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float p0 = ex2(sc[8 * k2 + e]), p1 = ex2(sc[8 * k2 + e + 1]);
                ls += p0 + p1;
                const f32x2 pp = {p0, p1};
                const half2v ph = __builtin_convertvector(pp, half2v);
                pf[k2][e] = ph[0];
                pf[k2][e + 1] = ph[1];
            }
        lsum[qb] += ls;
        asm volatile("" : "+v"(pf[0]), "+v"(pf[1]), "+v"(lsum[qb]));   // (P is wanted HERE: left alone, the compiler sinks the exponentials to P's first use, one region later)
    };
    auto pv = [&](const half8 (*vf)[2], const half8* pf, int qb) {
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) o[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[dt][k2], pf[k2], o[qb][dt], 0, 0, 0);
    };
#define A2_RDK(sub)  _Pragma("unroll") for (int kk = 0; kk < 4; kk++) GEO_DSR(kf[kk], ak[kk] + bo, (sub) * 4096)
#define A2_RDV(sub)                                                          \
    _Pragma("unroll") for (int dt = 0; dt < 2; dt++)                         \
        _Pragma("unroll") for (int k2 = 0; k2 < 2; k2++) GEO_DSR(vf[dt][k2], av[sub][k2] + bo, dt * 4096)
#define A2_WAITK() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]))
#define A2_WAITKV()                                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(vf[0][0]), "+v"(vf[0][1]), \
                 "+v"(vf[1][0]), "+v"(vf[1][1]))
#define A2_WAITV() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]))

    const int nt = L / AK;
    A2_ISSUE(0, 0);
    for (int t = 0; t < nt; t++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // tile t has landed for every wave, and everybody is done reading the other buffer
        if (t + 1 < nt) {
            if (t & 1) A2_ISSUE(t + 1, 0);
            else A2_ISSUE(t + 1, 1);
        }
        asm volatile("" ::: "memory");
        const unsigned bo = (unsigned)(t & 1) * (2u * AK * 8u * 16u);   // 16 KB per buffer
        half8 kf[4], vf[2][2], pfA[2], pfB[2];
        f32x16 s0, s1, s2, s3;
        // R0
        A2_RDK(0);
        A2_WAITK();
        s0 = qk(kf, 0);
        s1 = qk(kf, 1);
        __builtin_amdgcn_sched_barrier(0);
        A2_RDK(1);
        A2_RDV(0);
        // R1
        check(s0, 0, t == 0);
        A2_WAITKV();
        __builtin_amdgcn_sched_barrier(0);
        s2 = qk(kf, 0);
        expo(s0, 0, pfA);
        __builtin_amdgcn_sched_barrier(0);
        // R2
        check(s1, 1, t == 0);
        __builtin_amdgcn_sched_barrier(0);
        s3 = qk(kf, 1);
        pv(vf, pfA, 0);
        expo(s1, 1, pfB);
        __builtin_amdgcn_sched_barrier(0);
        // R3
        check(s2, 0, false);
        __builtin_amdgcn_sched_barrier(0);
        pv(vf, pfB, 1);
        expo(s2, 0, pfA);
        __builtin_amdgcn_sched_barrier(0);
        A2_RDV(1);
        // R4
        check(s3, 1, false);
        A2_WAITV();
        __builtin_amdgcn_sched_barrier(0);
        pv(vf, pfA, 0);
        expo(s3, 1, pfB);
        __builtin_amdgcn_sched_barrier(0);
        // R5
        pv(vf, pfB, 1);
    }

#undef A2_ISSUE
#undef A2_RDK
#undef A2_RDV
#undef A2_WAITK
#undef A2_WAITKV
#undef A2_WAITV
    // ---- normalise and store: the lane holds, for ONE query, d = 32 dt + 8 g + 4 hi + (0..3)
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
        const float lt = lsum[qb] + other_half(lsum[qb]);
        const float inv = 1.0f / lt;
        const int row = q0 + qb * 32 + l31;
        // MINUS the log2 of the softmax denominator in the scores' own (log2) units: the backward pass starts its score accumulators
        // at this value and gets P = exp2(s - lse) without a subtraction; -inf for the rows that pad the last tile of 64 (P = 0)
        if (nlse && hi == 0 && row < ((M + 63) & ~63)) nlse[(size_t)row * heads + head] = (row < M) ? negm[qb][0] - __builtin_amdgcn_logf(lt) : -INFINITY;
        if (lse_nat && hi == 0 && row < M) lse_nat[(size_t)head * M + row] = (__builtin_amdgcn_logf(lt) - negm[qb][0]) * 0.6931471805599453f;
        if (row < M) {
#pragma unroll
            for (int dt = 0; dt < 2; dt++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    half4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (h16)(o[qb][dt][4 * g + e] * inv);
                    *reinterpret_cast<half4*>(O + (size_t)row * ldo + head * 64 + dt * 32 + 8 * g + 4 * hi) = v;
                }
        }
    }
}


// hy3dgen's qk_norm on the key side: LayerNorm over the 64 dimensions of every head of K, in place (one thread per token
// and head; 3072 x 16 of them, once per set of latent tokens).  kn: gain (64), bias (64), eps.
__global__ void k_geo_knorm(h16* __restrict__ KV, int ldkv, int L, int heads, const float* __restrict__ kn) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L * heads) return;
    h16* k = KV + (size_t)(i / heads) * ldkv + (i % heads) * 64;
    float x[64], sm = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const half8 v = *reinterpret_cast<const half8*>(k + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            x[8 * c + e] = (float)v[e];
            sm += x[8 * c + e];
        }
    }
    const float mean = sm * (1.0f / 64.0f);
    float sq = 0.0f;
#pragma unroll
    for (int c = 0; c < 64; c++) {
        x[c] -= mean;
        sq += x[c] * x[c];
    }
    const float rstd = rsqrtf(sq * (1.0f / 64.0f) + kn[128]);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        half8 v;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (h16)(x[8 * c + e] * rstd * kn[8 * c + e] + kn[64 + 8 * c + e]);
        *reinterpret_cast<half8*>(k + 8 * c) = v;
    }
}

// V half of the KV projection (L rows, row stride ldkv, columns width + head*64 + d) -> Vt[head][d][pos(l)]: within every
// block of 16 keys, key kq goes to position (kq & 3) | ((kq & 4) << 1) | ((kq & 8) >> 1) (keys 4-7 and 8-11 swap places):
// the order in which a lane of the P^T fragment holds its 8 keys (C layout of the 32x32 MFMA: rows (r & 3) + 8 (r >> 2) + 4 hi).
__global__ __launch_bounds__(256) void k_geo_pack_vt(const h16* __restrict__ KV, int ldkv, int width, int L, h16* __restrict__ Vt, int col0 = -1, int hs = 64,
                                                     AttnB bs = AttnB{}) {
    KV += blockIdx.z * bs.a, Vt += blockIdx.z * bs.b;   // (bs: a = KV, b = Vt; the image is blockIdx.z here)
    // one thread per (column, block of 16 keys): sixteen strided reads (coalesced across the threads of a wave: consecutive columns),
    // one 32-byte row segment written in the permuted order.  (One thread per ELEMENT, 3072 x 1024 two-byte stores, took 19 us.)
    const int c = blockIdx.x * 256 + threadIdx.x;  // head * 64 + d
    const int l0 = blockIdx.y * 16;
    if (c >= width) return;
    // col0: first column of the half to transpose (default: the V half); hs: distance of two heads inside a row (64: side by side)
    const h16* src = KV + (size_t)l0 * ldkv + (col0 < 0 ? width : col0) + (c >> 6) * hs + (c & 63);
    half8 lo, hi8;
#pragma unroll
    for (int pos = 0; pos < 16; pos++) {
        const int kq = (pos & 3) | ((pos & 4) << 1) | ((pos & 8) >> 1);   // the permutation is an involution: position pos holds key kq
        const h16 v = src[(size_t)kq * ldkv];
        if (pos < 8) lo[pos] = v;
        else hi8[pos - 8] = v;
    }
    h16* dst = Vt + (size_t)c * L + l0;
    *reinterpret_cast<half8*>(dst) = lo;
    *reinterpret_cast<half8*>(dst + 8) = hi8;
}

// ------------------------------------------------------------------------------------------------
// Backward of the cross attention with respect to K and V (the query side is constant: the queries are grid points).
//   P = exp2(S - lse), S = Qs K^T (Qs = Q log2(e) / 8);  dV = P^T dO;  dP = dO V^T;  dS = P (dP - delta), delta = rowsum(dO O);
//   dK = dS^T Qs ln 2.
// Workgroup = 4 waves x 32 keys of ONE head (K and V fragments of a wave's keys stay in registers); it walks query tiles of 64
// (tile ti = split + n splits: `splits` workgroups share a key block and add their sums with float atomics).  Per tile the
// 4 waves share, in LDS: Qs and dO (rows = queries: A operands of S and dP) and their TRANSPOSED copies (rows = d, columns
// = queries in operand order, written by the GEMM epilogue EP_TRANS: A operands of dK^T = Qs^T dS and dV^T = dO^T P).  S and
// dP come out with the key in the lane and the query in the register index, which is exactly the B operand of those two
// products (k = query) -- no transpose, no cross-lane traffic; lse / delta of the register's query come from LDS.
// ------------------------------------------------------------------------------------------------
constexpr int BQ = 64;  // queries per tile

__device__ __forceinline__ void glds4(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ f32x16 cat16(f32x4 a, f32x4 b, f32x4 c, f32x4 d) {
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    const f32x8 lo = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}

// Staging: LDS-DMA (no staging registers, no ds_write pass), tile t + 1 in flight while tile t is computed; every LDS read of the
// loop is inline assembly with its own s_waitcnt (a read the compiler can see after an LDS-DMA gets a vmcnt(0) in front of it).
// The accumulators of S and dP are INITIALISED by reads of -lse / -delta (stored negated by the producers), so S - lse and
// dP - delta come out of the matrix pipe.  One tile = ten groups of four matrix instructions; the reads of group i + 1 are
// issued before the instructions of group i, and the softmax / dS arithmetic of one half of the queries is written BETWEEN the
// matrix instructions of the other half's groups: a wave issues in order, so vector work only overlaps its own matrix work when
// the two alternate in the instruction stream.
__global__ __launch_bounds__(256, 2) void k_geo_attn_bwd(const h16* __restrict__ Qs, const h16* __restrict__ QsT, const h16* __restrict__ dO,
                                                         const h16* __restrict__ dOT, int ldt, const float* __restrict__ nlse,
                                                         const float* __restrict__ ndelta, const h16* __restrict__ KV, int ldkv, int width,
                                                         int heads, int M, int splits, int L, int accumulate, float* __restrict__ part,
                                                         const int* __restrict__ Mdev, const h16* __restrict__ Vp = nullptr, int khs = 64,
                                                         h16* __restrict__ dk16 = nullptr, h16* __restrict__ dv16 = nullptr, int ld16 = 0, AttnB bs = AttnB{}) {
    // bs (with dk16 / dv16 only: the partial-sum route stays one image per launch): a = Qs, b = QsT, c = dO, d = dOT, e = nlse, f = ndelta,
    // g = KV and Vp, h = dk16 and dv16
    if (blockIdx.y) {
        Qs += blockIdx.y * bs.a, QsT += blockIdx.y * bs.b, dO += blockIdx.y * bs.c, dOT += blockIdx.y * bs.d, nlse += blockIdx.y * bs.e, ndelta += blockIdx.y * bs.f;
        KV += blockIdx.y * bs.g;
        if (Vp) Vp += blockIdx.y * bs.g;
        if (dk16) dk16 += blockIdx.y * bs.h, dv16 += blockIdx.y * bs.h;
    }
    // K rows at KV + key ldkv + head khs; V rows at Vp + ... (default: the V half of the decoder's K | V projection, KV + width)
    // dk16 / dv16 (with splits == 1: this workgroup's sums ARE the gradient): dK / dV as fp16 rows of stride ld16, heads side by side,
    // instead of the fp32 partial sums in `part` -- no reduction pass
    if (!Vp) Vp = KV + width;
    if (Mdev) M = max(min(M, *Mdev), 0);   // 0 rows: the first block of a call still writes its (zero) partial sums
    __shared__ uint4 lds[2][4][BQ * 8];   // [buffer][Qs | dO | Qs^T | dO^T][64 rows x 8 chunks] = 64 KB
    __shared__ float lsd[2][2][BQ];       // [buffer][-lse | -delta]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    // The L / 128 workgroups of one (head, split) GROUP stream the same query tiles: a group's workgroups get consecutive slots on
    // ONE XCD (workgroup b runs on XCD b % 8), so that a tile comes over the fabric once per group and out of that XCD's L2 for the
    // rest (dealt out round robin over the XCDs every tile was fetched by all of them: 463 -> 450 us).
    const int nkb = L / 128, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = (slot / nkb) * 8 + xcd, kt = slot % nkb;
    if (grp >= heads * splits) return;
    const int split = grp % splits, head = grp / splits;
    const int key = kt * 128 + w * 32 + l31;
    half8 kf[4], vf[4];   // B operands: lane (key, hi) holds K / V [key][16 kk + 8 hi .. + 7]
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        kf[kk] = *reinterpret_cast<const half8*>(KV + (size_t)key * ldkv + head * khs + 16 * kk + 8 * hi);
        vf[kk] = *reinterpret_cast<const half8*>(Vp + (size_t)key * ldkv + head * khs + 16 * kk + 8 * hi);
    }
    f32x16 dk0, dk1, dv0, dv1;  // dK^T / dV^T [d tile]: rows d, columns key
#pragma unroll
    for (int r = 0; r < 16; r++) dk0[r] = dk1[r] = dv0[r] = dv1[r] = 0.0f;
    const int ntiles = (M + BQ - 1) / BQ;

    // DMA sources: wave w stages rows 16 w .. 16 w + 15 of each of the four tiles, two pieces of 8 rows x 128 bytes; a lane's 16
    // bytes land at (row, slot = lane & 7), so it FETCHES chunk slot ^ swz(row)
    const int prow = lane >> 3, pslot = lane & 7;
    const int row0 = 16 * w + prow, row1 = row0 + 8;
    const int col0 = head * 64 + ((pslot ^ swz(row0)) << 3), col1 = head * 64 + ((pslot ^ swz(row1)) << 3);
    const size_t tr0 = (size_t)(head * 64 + row0) * ldt + ((pslot ^ swz(row0)) << 3), tr1 = (size_t)(head * 64 + row1) * ldt + ((pslot ^ swz(row1)) << 3);
#define BWD_ISSUE(ti, buf)                                                                   \
    do {                                                                                     \
        const int q0_ = (ti) * BQ;                                                           \
        const size_t ra_ = (size_t)min(q0_ + row0, M - 1) * width + col0;                    \
        const size_t rb_ = (size_t)min(q0_ + row1, M - 1) * width + col1;                    \
        glds16(Qs + ra_, &lds[buf][0][(16 * w) * 8]);                                        \
        glds16(Qs + rb_, &lds[buf][0][(16 * w + 8) * 8]);                                    \
        glds16(dO + ra_, &lds[buf][1][(16 * w) * 8]);                                        \
        glds16(dO + rb_, &lds[buf][1][(16 * w + 8) * 8]);                                    \
        glds16(QsT + tr0 + q0_, &lds[buf][2][(16 * w) * 8]);                                 \
        glds16(QsT + tr1 + q0_, &lds[buf][2][(16 * w + 8) * 8]);                             \
        glds16(dOT + tr0 + q0_, &lds[buf][3][(16 * w) * 8]);                                 \
        glds16(dOT + tr1 + q0_, &lds[buf][3][(16 * w + 8) * 8]);                             \
        if (w == 0) glds4(nlse + (size_t)(q0_ + lane) * heads + head, &lsd[buf][0][0]);      \
        if (w == 1) glds4(ndelta + (size_t)(q0_ + lane) * heads + head, &lsd[buf][1][0]);    \
    } while (0)

    // read addresses (buffer 0): A fragments of the row-major tiles (row = query), of the transposed tiles (row = d), -lse / -delta
    const unsigned lbase = lds_addr(&lds[0][0][0]);
    unsigned ar0[4], ar1[4], at00[2], at01[2], at10[2], at11[2];  // at<dt><qb>[ks]
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        ar0[kk] = lbase + l31 * 128 + (((2 * kk + hi) ^ swz(l31)) << 4);
        ar1[kk] = lbase + (32 + l31) * 128 + (((2 * kk + hi) ^ swz(32 + l31)) << 4);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        at00[ks] = lbase + l31 * 128 + (((2 * ks + hi) ^ swz(l31)) << 4);
        at01[ks] = lbase + l31 * 128 + (((4 + 2 * ks + hi) ^ swz(l31)) << 4);
        at10[ks] = lbase + (32 + l31) * 128 + (((2 * ks + hi) ^ swz(32 + l31)) << 4);
        at11[ks] = lbase + (32 + l31) * 128 + (((4 + 2 * ks + hi) ^ swz(32 + l31)) << 4);
    }
    const unsigned sbase = lds_addr(&lsd[0][0][0]) + hi * 16;

    // one group's fragments: Qs / dO (A operands of S / dP, K steps 0..3) or dO^T, Qs^T (A operands of dV / dK, query steps 0, 1)
#define BWD_RD_ROWS(F, AR, TILE_OFF)     \
    GEO_DSR(F##0, AR[0] + bo, TILE_OFF); \
    GEO_DSR(F##1, AR[1] + bo, TILE_OFF); \
    GEO_DSR(F##2, AR[2] + bo, TILE_OFF); \
    GEO_DSR(F##3, AR[3] + bo, TILE_OFF)
#define BWD_RD_T(F, AT)                \
    GEO_DSR(F##0, AT[0] + bo, 24576);  \
    GEO_DSR(F##1, AT[0] + bo, 16384);  \
    GEO_DSR(F##2, AT[1] + bo, 24576);  \
    GEO_DSR(F##3, AT[1] + bo, 16384)
    // accumulator initial values of query half QB: -lse into S##0..3, -delta into D##0..3 (register r <-> query 32 QB + (r & 3) + 8 (r >> 2) + 4 hi)
#define BWD_RD_INIT(QB, S, D)                      \
    GEO_DSR(S##0, sl, (QB) * 128 + 0);             \
    GEO_DSR(S##1, sl, (QB) * 128 + 32);            \
    GEO_DSR(S##2, sl, (QB) * 128 + 64);            \
    GEO_DSR(S##3, sl, (QB) * 128 + 96);            \
    GEO_DSR(D##0, sl, 256 + (QB) * 128 + 0);       \
    GEO_DSR(D##1, sl, 256 + (QB) * 128 + 32);      \
    GEO_DSR(D##2, sl, 256 + (QB) * 128 + 64);      \
    GEO_DSR(D##3, sl, 256 + (QB) * 128 + 96)
#define BWD_WAIT4(F) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F##0), "+v"(F##1), "+v"(F##2), "+v"(F##3))
#define BWD_WAIT12(F, S, D)                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                      \
                 : "+v"(F##0), "+v"(F##1), "+v"(F##2), "+v"(F##3), "+v"(S##0), "+v"(S##1), "+v"(S##2), "+v"(S##3), "+v"(D##0), "+v"(D##1), \
                   "+v"(D##2), "+v"(D##3))
#define BWD_MFMA(ACC, F, I, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(F##I, B, ACC, 0, 0, 0)
    // P (fp16) = exp2(S - lse) of registers [R0, R0 + 8) -> one B operand (k = query); likewise dS = P (dP - delta)
#define BWD_EXP8(PF, SC, R0)                                                       \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; e_++) PF[e_] = ex2(SC[(R0) + e_])
#define BWD_PACK8(PB, DB, PF, DPV, R0)                                                                               \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; e_ += 2) {                                                            \
        const f32x2 pp_ = {PF[e_], PF[e_ + 1]}, dd_ = {PF[e_] * DPV[(R0) + e_], PF[e_ + 1] * DPV[(R0) + e_ + 1]};    \
        const half2v ph_ = __builtin_convertvector(pp_, half2v), dh_ = __builtin_convertvector(dd_, half2v);         \
        PB[e_] = ph_[0], PB[e_ + 1] = ph_[1], DB[e_] = dh_[0], DB[e_ + 1] = dh_[1];                                  \
    }

    int ti = split, it = 0;
    if (ti < ntiles) BWD_ISSUE(ti, 0);
    for (; ti < ntiles; ti += splits, it++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // tile `it` has landed for every wave, and every wave is done reading the other buffer
        if (ti + splits < ntiles) {
            if (it & 1) BWD_ISSUE(ti + splits, 0);
            else BWD_ISSUE(ti + splits, 1);
        }
        asm volatile("" ::: "memory");
        const unsigned bo = (unsigned)(it & 1) << 15, sl = sbase + ((unsigned)(it & 1) << 9);
        half8 fa0, fa1, fa2, fa3, fb0, fb1, fb2, fb3;    // two fragment groups in flight
        f32x4 s00, s01, s02, s03, d00, d01, d02, d03, s10, s11, s12, s13, d10, d11, d12, d13;
        float pf[8], pg[8];
        half8 pb0[2], db0[2], pb1[2], db1[2];
        BWD_RD_INIT(0, s0, d0);
        BWD_RD_ROWS(fa, ar0, 0);                     // Qs, query half 0
        BWD_WAIT12(fa, s0, d0);
        f32x16 sc0 = cat16(s00, s01, s02, s03), dp0 = cat16(d00, d01, d02, d03);
        BWD_RD_ROWS(fb, ar0, 8192);                  // dO, half 0
        BWD_RD_INIT(1, s1, d1);
        // ---- group 1: S, half 0
        BWD_MFMA(sc0, fa, 0, kf[0]); BWD_MFMA(sc0, fa, 1, kf[1]); BWD_MFMA(sc0, fa, 2, kf[2]); BWD_MFMA(sc0, fa, 3, kf[3]);
        BWD_WAIT12(fb, s1, d1);
        f32x16 sc1 = cat16(s10, s11, s12, s13), dp1 = cat16(d10, d11, d12, d13);
        BWD_RD_ROWS(fa, ar1, 0);                     // Qs, half 1
        // ---- group 2: dP, half 0
        BWD_MFMA(dp0, fb, 0, vf[0]); BWD_MFMA(dp0, fb, 1, vf[1]); BWD_MFMA(dp0, fb, 2, vf[2]); BWD_MFMA(dp0, fb, 3, vf[3]);
        BWD_WAIT4(fa);
        BWD_RD_ROWS(fb, ar1, 8192);                  // dO, half 1
        // ---- group 3: S, half 1 || exp of half 0
        BWD_MFMA(sc1, fa, 0, kf[0]);
        BWD_EXP8(pf, sc0, 0);
        BWD_MFMA(sc1, fa, 1, kf[1]);
        BWD_MFMA(sc1, fa, 2, kf[2]);
        BWD_EXP8(pg, sc0, 8);
        BWD_MFMA(sc1, fa, 3, kf[3]);
        BWD_WAIT4(fb);
        BWD_RD_T(fa, at00);                          // dO^T / Qs^T rows d 0..31, half 0
        // ---- group 4: dP, half 1 || dS of half 0
        BWD_MFMA(dp1, fb, 0, vf[0]);
        BWD_PACK8(pb0[0], db0[0], pf, dp0, 0);
        BWD_MFMA(dp1, fb, 1, vf[1]);
        BWD_MFMA(dp1, fb, 2, vf[2]);
        BWD_PACK8(pb0[1], db0[1], pg, dp0, 8);
        BWD_MFMA(dp1, fb, 3, vf[3]);
        BWD_WAIT4(fa);
        BWD_RD_T(fb, at10);                          // rows d 32..63, half 0
        // ---- group 5: dV, dK (d 0..31) of half 0 || exp of half 1
        BWD_MFMA(dv0, fa, 0, pb0[0]);
        BWD_EXP8(pf, sc1, 0);
        BWD_MFMA(dk0, fa, 1, db0[0]);
        BWD_MFMA(dv0, fa, 2, pb0[1]);
        BWD_EXP8(pg, sc1, 8);
        BWD_MFMA(dk0, fa, 3, db0[1]);
        BWD_WAIT4(fb);
        BWD_RD_T(fa, at01);                          // rows d 0..31, half 1
        // ---- group 6: dV, dK (d 32..63) of half 0 || dS of half 1
        BWD_MFMA(dv1, fb, 0, pb0[0]);
        BWD_PACK8(pb1[0], db1[0], pf, dp1, 0);
        BWD_MFMA(dk1, fb, 1, db0[0]);
        BWD_MFMA(dv1, fb, 2, pb0[1]);
        BWD_PACK8(pb1[1], db1[1], pg, dp1, 8);
        BWD_MFMA(dk1, fb, 3, db0[1]);
        BWD_WAIT4(fa);
        BWD_RD_T(fb, at11);                          // rows d 32..63, half 1
        // ---- groups 7, 8: dV, dK of half 1
        BWD_MFMA(dv0, fa, 0, pb1[0]); BWD_MFMA(dk0, fa, 1, db1[0]); BWD_MFMA(dv0, fa, 2, pb1[1]); BWD_MFMA(dk0, fa, 3, db1[1]);
        BWD_WAIT4(fb);
        BWD_MFMA(dv1, fb, 0, pb1[0]); BWD_MFMA(dk1, fb, 1, db1[0]); BWD_MFMA(dv1, fb, 2, pb1[1]); BWD_MFMA(dk1, fb, 3, db1[1]);
    }
#undef BWD_ISSUE
#undef BWD_RD_ROWS
#undef BWD_RD_T
#undef BWD_RD_INIT
#undef BWD_WAIT4
#undef BWD_WAIT12
#undef BWD_MFMA
#undef BWD_EXP8
#undef BWD_PACK8
    __syncthreads();
    f32x16 dk[2] = {dk0, dk1}, dv[2] = {dv0, dv1};
    // This workgroup's sums -> part[split][key][dK (ln 2 folded in) | dV]: the (split, key block, head) slice is this workgroup's
    // alone, launches of successive row blocks are ordered by the stream, so a plain read-modify-write accumulates over the
    // row blocks -- no atomics (16 384 per workgroup were 0.8 of the kernel's time at 21-26 memory-side atomics per ns), and
    // the result does not depend on the order anything ran in.  Through LDS: a lane holds 4 consecutive d of ONE key.
    if (accumulate && split >= ntiles) return;   // nothing to add
    float* stage = reinterpret_cast<float*>(&lds[0][0][0]) + w * (32 * 68);
    float* gk = part + ((size_t)split * L + kt * 128 + w * 32) * (2 * width) + head * 64;
#pragma unroll
    for (int m = 0; m < 2; m++) {
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = m ? dv[dt][4 * g + e] : dk[dt][4 * g + e] * 0.6931471805599453f;
                *reinterpret_cast<f32x4*>(stage + l31 * 68 + dt * 32 + 8 * g + 4 * hi) = v;
            }
        __syncthreads();
#pragma unroll
        for (int it2 = 0; it2 < 8; it2++) {
            const int idx = it2 * 64 + lane, kr = idx >> 4, c4 = idx & 15;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + kr * 68 + c4 * 4);
            if (dk16) {
                half4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = (h16)v[e];
                *reinterpret_cast<half4*>((m ? dv16 : dk16) + (size_t)(kt * 128 + w * 32 + kr) * ld16 + head * 64 + c4 * 4) = o;
                continue;
            }
            float* dst = gk + (size_t)kr * (2 * width) + m * width + c4 * 4;
            if (accumulate) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] += o[e];
            }
            *reinterpret_cast<f32x4*>(dst) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ of an attention (round 5: the self-attention layers of the ShapeVAE transformer in front of the decoder -- latent2sdf runs and
// back-propagates sixteen of them in every inner iteration, PL:295, 1391-1393, 1507-1509; in the decoder's own cross attention the
// queries are grid points and have no gradient).   S = Qs K^T (log2 domain), P = exp2(S - lse), dP = dO V^T, dS = P (dP - delta),
// dQ = dS K / 8.  The mirror image of k_geo_attn: a workgroup = 4 waves x 32 queries of ONE head, the queries stay in the lanes
// (Qs and dO fragments in registers), 64-key tiles stream through LDS -- K rows and V rows (A operands of S^T = K Qs^T and
// dP^T = V dO^T: a lane owns a query, so -lse and -delta are the accumulators' initial values) and K^T (transposed, key-permuted
// like V^T in the forward: A operand of dQ^T += K^T dS^T, whose B operand is the dS^T accumulator itself, packed to fp16).
// Tiles by LDS-DMA (buffer_load ... lds), two buffers.
// ------------------------------------------------------------------------------------------------
constexpr int DQQ = 128;   // queries per workgroup of k_geo_attn_dq
__global__ __launch_bounds__(256, 2) void k_geo_attn_dq(const h16* __restrict__ Qs, const h16* __restrict__ dO, int ldq, const h16* __restrict__ KV,
                                                        int ldkv, int width, const h16* __restrict__ Kt, int L, const float* __restrict__ nlse,
                                                        const float* __restrict__ ndelta, h16* __restrict__ dQ, int M, int heads,
                                                        const h16* __restrict__ Vp = nullptr, int khs = 64, int lddq = 0, AttnB bs = AttnB{}) {
    // bs: a = Qs, b = dO, c = KV and Vp, d = Kt, e = nlse, f = ndelta, g = dQ
    if (blockIdx.y) {
        Qs += blockIdx.y * bs.a, dO += blockIdx.y * bs.b, KV += blockIdx.y * bs.c, Kt += blockIdx.y * bs.d, nlse += blockIdx.y * bs.e, ndelta += blockIdx.y * bs.f;
        dQ += blockIdx.y * bs.g;
        if (Vp) Vp += blockIdx.y * bs.c;
    }
    if (!Vp) Vp = KV + width;   // (K / V rows as in k_geo_attn_bwd)
    if (lddq == 0) lddq = ldq;  // row stride of dQ (default: that of Qs / dO)
    __shared__ uint4 lds[2][3][AK * 8];  // [buffer][K | V | K^T][64 rows x 8 chunks] = 48 KB
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int head, qblk;
    if ((heads & 7) == 0) {
        const int hpx = heads >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        head = xcd * hpx + j % hpx;
        qblk = j / hpx;
    } else {
        head = blockIdx.x % heads;
        qblk = blockIdx.x / heads;
    }
    if (qblk * DQQ >= M) return;
    const int q0 = qblk * DQQ + w * 32;   // 32 queries per wave: two query blocks' fragments + accumulators do not fit 256 registers

    half8 qf[1][4], dof[1][4];  // B operands: lane (q, hi) holds Qs / dO [q][16 kk + 8 hi .. + 7]
    float nl[1], nd[1];
#pragma unroll
    for (int qb = 0; qb < 1; qb++) {
        const int row = min(q0 + qb * 32 + l31, M - 1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            qf[qb][kk] = *reinterpret_cast<const half8*>(Qs + (size_t)row * ldq + head * 64 + 16 * kk + 8 * hi);
            dof[qb][kk] = *reinterpret_cast<const half8*>(dO + (size_t)row * ldq + head * 64 + 16 * kk + 8 * hi);
        }
        nl[qb] = nlse[(size_t)row * heads + head];
        nd[qb] = ndelta[(size_t)row * heads + head];
    }
    // DMA: per key tile every wave brings pieces 2 w and 2 w + 1 (8 rows x 128 bytes each) of each of the three tiles
    const int srow = lane >> 3, sslot = lane & 7;
    const int c0 = (sslot ^ ((srow >> 1) & 7)) << 3, c1 = (sslot ^ ((4 + (srow >> 1)) & 7)) << 3;
    const int vk0 = (srow * ldkv + c0) * 2, vk1 = (srow * ldkv + c1) * 2, vt0 = (srow * L + c0) * 2, vt1 = (srow * L + c1) * 2;
    const size_t kvspan = ((size_t)(L - 1) * ldkv + (size_t)(heads - 1) * khs + 64) * 2;   // bytes from the first to the last element of K (of V)
    const __amdgpu_buffer_rsrc_t rkv = __builtin_amdgcn_make_buffer_rsrc((void*)KV, (short)0, (int)min(kvspan, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rvv = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, (short)0, (int)min(kvspan, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rkt = __builtin_amdgcn_make_buffer_rsrc((void*)Kt, (short)0, (int)min((size_t)width * L * 2, (size_t)0x7fffffff), 0x00020000);
    const int r0 = 16 * w, r1 = 16 * w + 8;   // tile rows of this wave's two pieces
    const int sk = head * khs * 2, skt = head * 64 * L * 2;
#define DQ_ISSUE(t, buf)                                                                         \
    do {                                                                                         \
        const int kb_ = (t) * AK;                                                                \
        dma16(rkv, &lds[buf][0][r0 * 8], vk0, sk + (kb_ + r0) * ldkv * 2);                       \
        dma16(rkv, &lds[buf][0][r1 * 8], vk1, sk + (kb_ + r1) * ldkv * 2);                       \
        dma16(rvv, &lds[buf][1][r0 * 8], vk0, sk + (kb_ + r0) * ldkv * 2);                       \
        dma16(rvv, &lds[buf][1][r1 * 8], vk1, sk + (kb_ + r1) * ldkv * 2);                       \
        dma16(rkt, &lds[buf][2][r0 * 8], vt0, skt + (r0 * L + kb_) * 2);                         \
        dma16(rkt, &lds[buf][2][r1 * 8], vt1, skt + (r1 * L + kb_) * 2);                         \
    } while (0)

    f32x16 dq[1][2];  // [query block][d tile]: dQ^T, rows d, columns q
#pragma unroll
    for (int a = 0; a < 1; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) dq[a][b][r] = 0.0f;
    const unsigned lbase = lds_addr(&lds[0][0][0]);
    unsigned ak[4], at_[2][2];   // fragment addresses in buffer 0: K / V rows (row = key), K^T rows (row = d)
#pragma unroll
    for (int kk = 0; kk < 4; kk++) ak[kk] = lbase + l31 * 128 + (((2 * kk + hi) ^ swz(l31)) << 4);
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
        for (int k2 = 0; k2 < 2; k2++) at_[sub][k2] = lbase + l31 * 128 + (((4 * sub + 2 * k2 + hi) ^ swz(l31)) << 4);

    const int nt = L / AK;
    DQ_ISSUE(0, 0);
    for (int t = 0; t < nt; t++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // tile t has landed for every wave, and everybody is done reading the other buffer
        if (t + 1 < nt) {
            if (t & 1) DQ_ISSUE(t + 1, 0);
            else DQ_ISSUE(t + 1, 1);
        }
        asm volatile("" ::: "memory");
        const unsigned bo = (unsigned)(t & 1) * (3u * AK * 8u * 16u);   // 24 KB per buffer
#define DQ_SUB(SUB)                                                                                                                  \
        do {                                                                                                                         \
            half8 kf[4], vf[4], ktf[2][2];                                                                                           \
            _Pragma("unroll") for (int kk = 0; kk < 4; kk++) {                                                                       \
                GEO_DSR(kf[kk], ak[kk] + bo, (SUB) * 4096);          /* rows SUB * 32 + l31 (same swizzle: 32 rows = 4 x 8) */          \
                GEO_DSR(vf[kk], ak[kk] + bo, 8192 + (SUB) * 4096);                                                                   \
            }                                                                                                                        \
            _Pragma("unroll") for (int k2 = 0; k2 < 2; k2++) {                                                                       \
                GEO_DSR(ktf[0][k2], at_[SUB][k2] + bo, 16384);                                                                       \
                GEO_DSR(ktf[1][k2], at_[SUB][k2] + bo, 16384 + 4096);                                                                \
            }                                                                                                                        \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                                      \
                         : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]),      \
                           "+v"(ktf[0][0]), "+v"(ktf[0][1]), "+v"(ktf[1][0]), "+v"(ktf[1][1]));                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                       \
            _Pragma("unroll") for (int qb = 0; qb < 1; qb++) {                                                                       \
                f32x16 sc, dp;                                                                                                       \
                _Pragma("unroll") for (int r = 0; r < 16; r++) sc[r] = nl[qb], dp[r] = nd[qb];                                       \
                _Pragma("unroll") for (int kk = 0; kk < 4; kk++) sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kk], qf[qb][kk], sc, 0, 0, 0);   \
                _Pragma("unroll") for (int kk = 0; kk < 4; kk++) dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kk], dof[qb][kk], dp, 0, 0, 0);  \
                half8 dsf[2];                                                                                                        \
                _Pragma("unroll") for (int k2 = 0; k2 < 2; k2++)                                                                     \
                    _Pragma("unroll") for (int e = 0; e < 8; e += 2) {                                                               \
                        const f32x2 dd = {ex2(sc[8 * k2 + e]) * dp[8 * k2 + e], ex2(sc[8 * k2 + e + 1]) * dp[8 * k2 + e + 1]};        \
                        const half2v dh = __builtin_convertvector(dd, half2v);                                                       \
                        dsf[k2][e] = dh[0], dsf[k2][e + 1] = dh[1];                                                                  \
                    }                                                                                                                \
                _Pragma("unroll") for (int dt = 0; dt < 2; dt++)                                                                     \
                    _Pragma("unroll") for (int k2 = 0; k2 < 2; k2++)                                                                 \
                        dq[qb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ktf[dt][k2], dsf[k2], dq[qb][dt], 0, 0, 0);              \
                __builtin_amdgcn_sched_barrier(0);   /* one query block at a time: two sets of scores do not fit the register file */    \
            }                                                                                                                        \
        } while (0)
        DQ_SUB(0);
        DQ_SUB(1);
#undef DQ_SUB
    }
#undef DQ_ISSUE
    // ---- store: the lane holds, for ONE query, d = 32 dt + 8 g + 4 hi + (0..3); dQ = dS K / 8
#pragma unroll
    for (int qb = 0; qb < 1; qb++) {
        const int row = q0 + qb * 32 + l31;
        if (row < M) {
#pragma unroll
            for (int dt = 0; dt < 2; dt++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    half4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = (h16)(dq[qb][dt][4 * g + e] * 0.125f);
                    *reinterpret_cast<half4*>(dQ + (size_t)row * lddq + head * 64 + dt * 32 + 8 * g + 4 * hi) = v;
                }
        }
    }
}

// X (M rows, W columns) -> XT[c][pos(m)] (row length ldt), pos = m with bits 2 and 3 of (m & 15) swapped: the transposed copies
// k_geo_attn_bwd reads (what the GEMM epilogue EP_TRANS writes inside the decoder); columns beyond M are left alone (cleared by the caller)
// hs: distance of two heads inside a row of X; scale != 1: the values are multiplied (and rounded to fp16 again) first, and Xs (M x W,
// heads side by side) receives the scaled rows as well -- the scaled copy of Q the backward attention kernels stream
__global__ __launch_bounds__(256) void k_geo_transpose_perm(const h16* __restrict__ X, int ldx, int M, int W, h16* __restrict__ XT, int ldt, int hs = 64,
                                                            float scale = 1.0f, h16* __restrict__ Xs = nullptr, AttnB bs = AttnB{}) {
    X += blockIdx.y * bs.a, XT += blockIdx.y * bs.b;   // (bs: a = X, b = XT -- a column offset inside the batch's transposed image --, c = Xs)
    if (Xs) Xs += blockIdx.y * bs.c;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int m = i % M, c8 = i / M;
    if (c8 * 8 >= W) return;
    half8 v = *reinterpret_cast<const half8*>(X + (size_t)m * ldx + (c8 >> 3) * hs + (c8 & 7) * 8);
    if (scale != 1.0f)
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (h16)((float)v[e] * scale);
    if (Xs) *reinterpret_cast<half8*>(Xs + (size_t)m * W + c8 * 8) = v;
    const int pm = (m & ~15) | (m & 3) | ((m & 4) << 1) | ((m & 8) >> 1);
#pragma unroll
    for (int e = 0; e < 8; e++) XT[(size_t)(c8 * 8 + e) * ldt + pm] = v[e];
}

// grad_kv = sum over the splits of k_geo_attn_bwd's partial sums (fixed order: bitwise repeatable)
__global__ __launch_bounds__(256) void k_geo_dkv_reduce(const float* __restrict__ part, int splits, size_t n4, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = reinterpret_cast<const f32x4*>(part)[i];
    for (int sp = 1; sp < splits; sp++) {
        const f32x4 b = reinterpret_cast<const f32x4*>(part)[(size_t)sp * n4 + i];
#pragma unroll
        for (int e = 0; e < 4; e++) a[e] += b[e];
    }
    reinterpret_cast<f32x4*>(out)[i] = a;
}

// ... the same sum, handed out as two fp16 matrices (L x W each): dK and dV of foho_sdpa_bwd
__global__ __launch_bounds__(256) void k_geo_dkv_reduce16(const float* __restrict__ part, int splits, int L, int W, h16* __restrict__ dk, h16* __restrict__ dv,
                                                          int ldo = 0) {
    if (ldo == 0) ldo = W;   // row stride of dk / dv (the ShapeVAE transformer writes them into its (tokens, 3 W) gradient of q | k | v)
    const size_t n4 = (size_t)L * 2 * W / 4, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 a = reinterpret_cast<const f32x4*>(part)[i];
    for (int sp = 1; sp < splits; sp++) {
        const f32x4 b = reinterpret_cast<const f32x4*>(part)[(size_t)sp * n4 + i];
#pragma unroll
        for (int e = 0; e < 4; e++) a[e] += b[e];
    }
    const size_t l = i / (W / 2), c = (i % (W / 2)) * 4;   // row, column within the 2 W wide row
    half4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = (h16)a[e];
    *reinterpret_cast<half4*>((c < (size_t)W ? dk + l * ldo + c : dv + l * ldo + (c - W))) = o;
}

// ndelta[q][head] = - sum_d dO[q][head, d] O[q][head, d] (negated: the backward attention starts its dP accumulators there); one
// wave per row, 8 lanes per head and half row; rows M .. the next multiple of 64 (the padding of the last query tile) get 0
__global__ __launch_bounds__(256) void k_geo_delta(const h16* __restrict__ dO, const h16* __restrict__ O, int width, int heads, int M,
                                                   float* __restrict__ ndelta, const int* __restrict__ Mdev, AttnB bs = AttnB{}) {
    dO += blockIdx.y * bs.a, O += blockIdx.y * bs.b, ndelta += blockIdx.y * bs.c;   // (bs: a = dO, b = O, c = ndelta)
    if (Mdev) M = min(M, *Mdev);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= ((M + 63) & ~63)) return;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int col = (c * 64 + lane) * 8;
        float s = 0.0f;
        if (col < width && row < M) {
            const half8 a = *reinterpret_cast<const half8*>(dO + (size_t)row * width + col), b = *reinterpret_cast<const half8*>(O + (size_t)row * width + col);
#pragma unroll
            for (int e = 0; e < 8; e++) s += (float)a[e] * (float)b[e];
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if ((lane & 7) == 0 && col < width) ndelta[(size_t)row * heads + col / 64] = -s;
    }
}

// Backward of a LayerNorm row (one wave per row): x = the forward's input, y = xhat gamma + beta.
// MODE 0: dy from DY (fp16), DX = (DR ? DR : 0) + dx.   MODE 1 (ln_post + output_proj): dy_i = gvec[row] w_out[i].
template <int MODE>
__global__ __launch_bounds__(256) void k_geo_ln_bwd(const h16* __restrict__ X, const float* __restrict__ gamma, const h16* __restrict__ DY,
                                                    const h16* __restrict__ DR, const float* __restrict__ gvec, float gain,
                                                    const float* __restrict__ w_out, h16* __restrict__ DX, int M, int width, float eps,
                                                    const int* __restrict__ Mdev) {
    if (Mdev) M = min(M, *Mdev);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float x[2][8], dy[2][8];
    float sum = 0.0f;
    const float gr = (MODE == 1) ? gvec[row] * gain : 0.0f;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int col = (c * 64 + lane) * 8;
        if (col < width) {
            const half8 h = *reinterpret_cast<const half8*>(X + (size_t)row * width + col);
            half8 g;
            if (MODE == 0) g = *reinterpret_cast<const half8*>(DY + (size_t)row * width + col);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                x[c][e] = (float)h[e];
                sum += x[c][e];
                dy[c][e] = ((MODE == 0) ? (float)g[e] : gr * w_out[col + e]) * (gamma ? gamma[col + e] : 1.0f);   // d / d xhat (gamma == NULL: all ones)
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) x[c][e] = dy[c][e] = 0.0f;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)width;
    float sq = 0.0f;
#pragma unroll
    for (int c = 0; c < 2; c++)
        if ((c * 64 + lane) * 8 < width)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float d = x[c][e] - mean;
                sq += d * d;
            }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq / (float)width + eps);
    float a = 0.0f, b = 0.0f;   // mean(d xhat), mean(d xhat . xhat)
#pragma unroll
    for (int c = 0; c < 2; c++)
        if ((c * 64 + lane) * 8 < width)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                a += dy[c][e];
                b += dy[c][e] * (x[c][e] - mean) * rstd;
            }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    a /= (float)width;
    b /= (float)width;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int col = (c * 64 + lane) * 8;
        if (col < width) {
            half8 r;
            if (DR) r = *reinterpret_cast<const half8*>(DR + (size_t)row * width + col);
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float dx = rstd * (dy[c][e] - a - (x[c][e] - mean) * rstd * b);
                o[e] = (h16)(dx + (DR ? (float)r[e] : 0.0f));
            }
            *reinterpret_cast<half8*>(DX + (size_t)row * width + col) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over rows of `width` <= 1024 halfs (one wave per row, fp32 statistics like torch's fp16 LayerNorm).
// MODE 0: Y = LN(X) gamma + beta (fp16).  MODE 1: logits[row] = prior(query) + gain * (LN(X) gamma + beta) . w_out + b_out.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_geo_ln(const h16* __restrict__ X, int ldx, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, h16* __restrict__ Y, int ldy, int M, int width,
                                                float eps, const float* __restrict__ w_out, float b_out, const float* __restrict__ queries,
                                                float radius, float sharpness, float gain, float* __restrict__ logits,
                                                const int* __restrict__ Mdev) {
    if (Mdev) M = min(M, *Mdev);
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[2][8];
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int col = (c * 64 + lane) * 8;
        if (col < width) {
            const half8 h = *reinterpret_cast<const half8*>(X + (size_t)row * ldx + col);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                v[c][e] = (float)h[e];
                sum += v[c][e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[c][e] = 0.0f;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)width;
    float sq = 0.0f;
#pragma unroll
    for (int c = 0; c < 2; c++)
        if ((c * 64 + lane) * 8 < width)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float d = v[c][e] - mean;
                sq += d * d;
            }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq / (float)width + eps);
    float dot = 0.0f;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int col = (c * 64 + lane) * 8;
        if (col < width) {
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float y = (v[c][e] - mean) * rstd * gamma[col + e] + beta[col + e];
                h[e] = (h16)y;
                if (MODE == 1) dot += (float)h[e] * w_out[col + e];  // the reference's LayerNorm output is fp16 before the last Linear
            }
            if (MODE == 0) *reinterpret_cast<half8*>(Y + (size_t)row * ldy + col) = h;
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
        if (lane == 0) {
            float learned = (float)(h16)(dot + b_out);
            float prior = 0.0f;
            if (sharpness != 0.0f) {
                const float x = queries[3 * (size_t)row], y = queries[3 * (size_t)row + 1], z = queries[3 * (size_t)row + 2];
                prior = (radius - sqrtf(x * x + y * y + z * z)) * sharpness;
            }
            logits[row] = prior + gain * learned;
        }
    }
}

// ---- LayerNorm folded into its neighbours (forward chain): the small kernels around the GEMM epilogues EP_STATS / EP_PREAFF / EP_LOGIT.
// Merge of a row's per-part statistics (parts of 64 columns each: mean_i, M2_i = sum of squared deviations from mean_i):
// mean = avg(mean_i), M2 = sum M2_i + 64 sum (mean_i - mean)^2 -- exact, no cancellation.
// Sixteen lanes per row (one per part; lanes beyond nslots idle), four rows per wave: a row's parts are contiguous.
__device__ __forceinline__ float sum16(float x) {   // over the 16 lanes of a DPP row; every lane gets it
    x = sum8(x);
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, true));   // row_mirror
}
__device__ __forceinline__ void merge_parts(float mean_i, float m2_i, bool live, int nslots, float& mean, float& var) {
    mean = sum16(live ? mean_i : 0.0f) / (float)nslots;
    const float d = mean_i - mean;
    var = sum16(live ? m2_i + 64.0f * d * d : 0.0f) / (64.0f * (float)nslots);
}
// out[row] = (rstd, rstd * mean) of the row whose parts c_proj's epilogue left in `stats`
__global__ __launch_bounds__(256) void k_geo_rowstat_finish(const float* __restrict__ stats, int nslots, int M, float eps, float2* __restrict__ out,
                                                            const int* __restrict__ Mdev) {
    if (Mdev) M = min(M, *Mdev);
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4), part = threadIdx.x & 15;
    const bool live = row < M && part < nslots;
    float2 st = float2{0.0f, 0.0f};
    if (live) st = reinterpret_cast<const float2*>(stats)[(size_t)row * nslots + part];
    float mean, var;
    merge_parts(st.x, st.y, live, nslots, mean, var);
    if (live && part == 0) {
        const float rstd = rsqrtf(var + eps);
        out[row] = float2{rstd, rstd * mean};
    }
}
// logits[row] = prior(query) + gain * half(rstd (sum x gw - mean GW) + C0 + b_out): ln_post + output_proj of the row whose parts fc2's
// epilogue left in `stats` (mean, M2, sum x gw per part); fold = (GW = sum gamma w_out, C0 = sum beta w_out)
__global__ __launch_bounds__(256) void k_geo_logit_finish(const float* __restrict__ stats, int nslots, int M, float eps, const float* __restrict__ fold,
                                                          float b_out, const float* __restrict__ queries, float radius, float sharpness, float gain,
                                                          float* __restrict__ logits, const int* __restrict__ Mdev) {
    if (Mdev) M = min(M, *Mdev);
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4), part = threadIdx.x & 15;
    const bool live = row < M && part < nslots;
    f32x4 st = {0.0f, 0.0f, 0.0f, 0.0f};
    if (live) st = reinterpret_cast<const f32x4*>(stats)[(size_t)row * nslots + part];
    float mean, var;
    merge_parts(st[0], st[1], live, nslots, mean, var);
    const float dot = sum16(st[2]);
    if (live && part == 0) {
        const float rstd = rsqrtf(var + eps);
        const float learned = (float)(h16)(rstd * (dot - mean * fold[0]) + fold[1] + b_out);
        float prior = 0.0f;
        if (sharpness != 0.0f) {
            const float x = queries[3 * (size_t)row], y = queries[3 * (size_t)row + 1], z = queries[3 * (size_t)row + 2];
            prior = (radius - sqrtf(x * x + y * y + z * z)) * sharpness;
        }
        logits[row] = prior + gain * learned;
    }
}
// The folded operands, once per prepare (one wave per fc1 output column c): W1f[c][k] = half(W1[c][k] gamma2[k]),
// fold1 = [b1'[c] = b1[c] + sum_k W1[c][k] beta2[k] | s[c] = sum_k W1f[c][k] (of the ROUNDED products: a constant row cancels exactly)].
__global__ __launch_bounds__(256) void k_geo_fold_fc1(const h16* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int F, int width, h16* __restrict__ W1f, float* __restrict__ fold1) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= F) return;
    float sb = 0.0f, ss = 0.0f;
    for (int k0 = lane * 8; k0 < width; k0 += 512) {
        const half8 wv = *reinterpret_cast<const half8*>(W1 + (size_t)c * width + k0);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float x = (float)wv[e];
            o[e] = (h16)(x * gamma[k0 + e]);
            ss += (float)o[e];
            sb = __builtin_fmaf(x, beta[k0 + e], sb);
        }
        *reinterpret_cast<half8*>(W1f + (size_t)c * width + k0) = o;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sb += __shfl_xor(sb, off);
        ss += __shfl_xor(ss, off);
    }
    if (lane == 0) {
        fold1[c] = b1[c] + sb;
        fold1[F + c] = ss;
    }
}
// fold2 = [b2 (width) | gw = gamma_post w_out (width) | GW = sum gw | C0 = sum beta_post w_out]; one workgroup
__global__ __launch_bounds__(256) void k_geo_fold_post(const float* __restrict__ b2, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ w_out, int width, float* __restrict__ fold2) {
    __shared__ float red[2][4];
    float g = 0.0f, c0 = 0.0f;
    for (int k = threadIdx.x; k < width; k += 256) {
        const float gw = gamma[k] * w_out[k];
        fold2[k] = b2[k];
        fold2[width + k] = gw;
        g += gw;
        c0 = __builtin_fmaf(beta[k], w_out[k], c0);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        g += __shfl_xor(g, off);
        c0 += __shfl_xor(c0, off);
    }
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = g, red[1][threadIdx.x >> 6] = c0;
    __syncthreads();
    if (threadIdx.x == 0) {
        fold2[2 * width] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        fold2[2 * width + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// Fourier embedding of the query points, [x, sin(x f_j), cos(x f_j)] flattened as (coordinate, frequency) like the
// reference's FourierEmbedder, rounded to fp16 and zero-padded to 64 columns (the K of the query projection GEMM).
__global__ __launch_bounds__(256) void k_geo_embed(const float* __restrict__ queries, int M, int n_freqs, const float* __restrict__ freqs,
                                                   h16* __restrict__ E, const int* __restrict__ Mdev) {
    if (Mdev) M = min(M, *Mdev);
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int row = gid >> 3, ch = gid & 7;
    if (row >= M) return;
    const float p[3] = {queries[3 * (size_t)row], queries[3 * (size_t)row + 1], queries[3 * (size_t)row + 2]};
    const int nf3 = 3 * n_freqs;
    half8 out;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int col = ch * 8 + e;
        float v = 0.0f;
        if (col < 3) v = p[col];
        else if (col < 3 + nf3) {
            const int i = col - 3;
            v = sinf(p[i / n_freqs] * freqs[i % n_freqs]);
        } else if (col < 3 + 2 * nf3) {
            const int i = col - 3 - nf3;
            v = cosf(p[i / n_freqs] * freqs[i % n_freqs]);
        }
        out[e] = (h16)v;
    }
    *reinterpret_cast<half8*>(E + (size_t)row * 64 + ch * 8) = out;
}

// ------------------------------------------------------------------------------------------------
// Compaction of the rows whose logit gradient is not zero (foho_geo_decode_bwd_rows), in row order: k_geo_rows_count leaves the number
// of such rows of every 2048-row segment, k_geo_rows_compact turns them into offsets (every workgroup sums the counts in front of it:
// at most a few hundred values) and writes the survivors' index, query point and gradient; the last workgroup publishes the total,
// the rows dropped for lack of capacity and the row count of every row block of the chain (the kernels' Mdev).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_sum_256(int v, int* sh) {   // sum over the 256 threads, known to all of them
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void k_geo_rows_count(const float* __restrict__ g, int64_t n, int* __restrict__ counts) {
    __shared__ int sh[4];
    const int64_t r0 = (int64_t)blockIdx.x * 2048 + threadIdx.x * 8;
    int c = 0;
#pragma unroll
    for (int e = 0; e < 8; e++)
        if (r0 + e < n && g[r0 + e] != 0.0f) c++;
    c = block_sum_256(c, sh);
    if (threadIdx.x == 0) counts[blockIdx.x] = c;
}
__global__ __launch_bounds__(256) void k_geo_rows_compact(const float* __restrict__ g, const float* __restrict__ queries, int64_t n, const int* __restrict__ counts,
                                                          int nscan, int64_t cap, int chunk, int nblk, int* __restrict__ mblk, float* __restrict__ q_out,
                                                          float* __restrict__ g_out, int* __restrict__ idx_out, int* __restrict__ stats_out) {
    __shared__ int sh[4], wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int before = 0;
    for (int i = tid; i < (int)blockIdx.x; i += 256) before += counts[i];
    before = block_sum_256(before, sh);
    const int64_t r0 = (int64_t)blockIdx.x * 2048 + tid * 8;
    float gv[8];
    int c = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        gv[e] = (r0 + e < n) ? g[r0 + e] : 0.0f;
        c += gv[e] != 0.0f;
    }
    int incl = c;   // inclusive scan over the wave, then over the four waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int pos = before + incl - c;
    for (int i = 0; i < wv; i++) pos += wsum[i];
#pragma unroll
    for (int e = 0; e < 8; e++)
        if (gv[e] != 0.0f) {
            if (pos < cap) {
                const int64_t r = r0 + e;
                idx_out[pos] = (int)r;
                g_out[pos] = gv[e];
                q_out[3 * (size_t)pos] = queries[3 * r], q_out[3 * (size_t)pos + 1] = queries[3 * r + 1], q_out[3 * (size_t)pos + 2] = queries[3 * r + 2];
            }
            pos++;
        }
    if ((int)blockIdx.x == nscan - 1 || nscan == 0) {
        const int total = before + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const int kept = (int)min((int64_t)total, cap);
        for (int cblk = tid; cblk < nblk; cblk += 256) mblk[2 + cblk] = max(min(kept - cblk * chunk, chunk), 0);
        if (tid == 0) {
            mblk[0] = total, mblk[1] = total - kept;
            if (stats_out) stats_out[0] = total, stats_out[1] = total - kept;
        }
    }
}
__global__ __launch_bounds__(256) void k_geo_zero16(uint4* __restrict__ p, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------ host side
// eps of one of the chain's LayerNorms: its own field when set, else ln_eps (callers of ABI 103 filled only that one)
static inline float eps_of(const foho_geo_weights* w, float own) { return own > 0.0f ? own : w->ln_eps; }
static bool launch_ok(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return false;
    }
    return true;
}

// Which kernel a GEMM runs on.  GV_AUTO: by shape (gemm() below); the others are for the unit entry point foho_geo_gemm (tests, A/B
// measurements) -- an ARGUMENT of the call, no process state: the library is driven from several threads (MeshGuidanceRunner, call_batch).
enum {
    GV_AUTO = 0,
    GV_128 = 1,         // k_geo_gemm: 128 x 128 tiles, two stages
    GV_LOCKSTEP = 2,    // k_geo_gemm256
    GV_PHASED = 3,      // k_geo_gemm8p, 256 x 256 tiles
    GV_DEEP = 4,        // k_geo_gemm_d4: 128 x 128 tiles, four-deep ring
    GV_PC = 5,          // k_geo_gemm_pc: the same with fill waves and matrix waves
    GV_PHASED192 = 6    // k_geo_gemm8p on 192 x 256 tiles
};
static unsigned cu_count() {   // a multiple of 8: the tile order deals consecutive tiles to the 8 XCDs
    static const unsigned ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        return (unsigned)(n & ~7);
    }();
    return ncu;
}
template <int EP>
static void launch_gemm(int variant, dim3 grid, hipStream_t s, const h16* A, int lda, const h16* Wt, int ldw, const float* bias, const h16* R, int ldr,
                        h16* C, int ldc, int M, int N, int K, float scale, h16* C2, int ldc2, const int* Mdev, const EpiAux& aux) {
    if (variant == GV_PHASED192)
        hipLaunchKernelGGL((k_geo_gemm8p<EP, 192>), dim3(std::min(grid.x, cu_count())), dim3(512), 0, s, A, lda, Wt, ldw, bias, R, ldr, C, ldc, M, N, K, scale, C2, ldc2, Mdev, aux);
    else if (variant == GV_PHASED)   // persistent: one workgroup per CU walks the tiles
        hipLaunchKernelGGL(k_geo_gemm8p<EP>, dim3(std::min(grid.x, cu_count())), dim3(512), 0, s, A, lda, Wt, ldw, bias, R, ldr, C, ldc, M, N, K, scale, C2, ldc2, Mdev, aux);
    else if (variant == GV_PC) hipLaunchKernelGGL(k_geo_gemm_pc<EP>, grid, dim3(512), 0, s, A, lda, Wt, ldw, bias, R, ldr, C, ldc, M, N, K, scale, C2, ldc2, Mdev, aux);
    else if (variant == GV_DEEP) hipLaunchKernelGGL(k_geo_gemm_d4<EP>, grid, dim3(256), 0, s, A, lda, Wt, ldw, bias, R, ldr, C, ldc, M, N, K, scale, C2, ldc2, Mdev, aux);
    else if (variant == GV_LOCKSTEP) hipLaunchKernelGGL(k_geo_gemm256<EP>, grid, dim3(512), 0, s, A, lda, Wt, ldw, bias, R, ldr, C, ldc, M, N, K, scale, C2, ldc2, Mdev, aux);
    else hipLaunchKernelGGL(k_geo_gemm<EP>, grid, dim3(256), 0, s, A, lda, Wt, ldw, bias, R, ldr, C, ldc, M, N, K, scale, C2, ldc2, Mdev, aux);
}

// ep: bit mask of EP_*.  R: the residual (EP_RESID) or the saved pre-activation (EP_GELUBWD); C2: the pre-activation output
// (EP_SAVEZ, leading dimension ldc2) or the transposed copy (EP_TRANS, row length ldc2).  Mdev: optional DEVICE row count (the
// launch is sized for M, the kernel works on min(M, *Mdev) rows).  variant: GV_*.
static int gemm(int ep, const h16* A, int lda, const h16* Wt, int ldw, const float* bias, const h16* R, int ldr, h16* C, int ldc, int M,
                int N, int K, float scale, hipStream_t s, h16* C2 = nullptr, int ldc2 = 0, const int* Mdev = nullptr, int variant = GV_AUTO,
                EpiAux aux = EpiAux{}) {
    if (M <= 0) return FOHO_OK;
    if (N % GN || K % GK || (lda & 7) || (ldw & 7) || (ldc & 7) || (R && !(ep & (EP_QNORM | EP_PREAFF)) && (ldr & 7))) return fail(FOHO_ERR_BAD_ARG, "geo gemm: N % 128, K % 64, leading dimensions % 8");
    if ((ep & (EP_RESID | EP_GELUBWD | EP_QNORM | EP_PREAFF)) && !R) return fail(FOHO_ERR_BAD_ARG, "geo gemm: epilogue operand missing");
    if ((ep & (EP_SAVEZ | EP_TRANS | EP_STATS | EP_LOGIT)) && !C2) return fail(FOHO_ERR_BAD_ARG, "geo gemm: second output missing");
    if (((ep & EP_PREAFF) && ldr != N) || ((ep & (EP_STATS | EP_LOGIT)) && ldc2 * 64 != N) || ((ep & EP_QKN) && (N % 192 || !(ep & EP_PREAFF))))
        return fail(FOHO_ERR_BAD_ARG, "geo gemm: folded-LayerNorm epilogue operands");
    const bool can_big = N % HN == 0 && K >= 256;
    // the phased kernel addresses its operands through 32-bit buffer offsets
    const bool can_phased = can_big && K / GK >= 2 && (size_t)M * lda * 2 < ((size_t)1 << 31) && (size_t)N * ldw * 2 < ((size_t)1 << 31);
    // the big GEMMs of the chain: 256 x 256 tiles -- when there are enough of them to occupy half the chip.  The ShapeVAE transformer's
    // N = 1024 products at M = 3072 are 48 such tiles on 256 CUs: 21 / 66 / 50 us at K = 1024 / 4096 / 3072 against 13 / 43 / 35 on 192 tiles of
    // 128 x 128 (scripts/dev/vae_bench.py --gemms)
    const long tiles256 = (long)((M + HM - 1) / HM) * (N / HN);
    const bool auto_choice = variant == GV_AUTO;
    if (variant == GV_AUTO) variant = (can_big && M >= 2048 && tiles256 >= 128) ? (can_phased ? GV_PHASED : GV_LOCKSTEP) : GV_128;
    if ((variant == GV_PHASED || variant == GV_PHASED192) && !can_phased) variant = can_big ? GV_LOCKSTEP : GV_128;
    // 256-row tiles whose last round leaves CUs idle (the transformer's M = 3072: 12 x 16 tiles on 256 CUs; four images' q | k | v: 2.25 rounds) ->
    // 192-row tiles when the busiest CU then multiplies fewer rows: rounds x rows per tile, 192-row rounds counted 8 % dearer (their phases 1 and 2
    // run four matrix instructions behind the same two barriers; the decoder's 49 152-row blocks are exact rounds either way and stay)
    if (variant == GV_PHASED && auto_choice && M > 192 && !((ep & EP_PACK) && aux.lt % 192)) {
        const long cu = (long)cu_count();
        const long t256 = 8L * (((M + HM - 1) / HM + 7) / 8) * (N / HN), t192 = 8L * (((M + 191) / 192 + 7) / 8) * (N / HN);
        if (((t192 + cu - 1) / cu) * 192 * 108 < ((t256 + cu - 1) / cu) * 256 * 100) variant = GV_PHASED192;
    }
    if (variant == GV_LOCKSTEP && !can_big) variant = GV_128;
    // 128 x 128 tiles that do not even fill the chip once: one workgroup per CU, and with one wave per SIMD the ring's fill and the matrix
    // work add up (NOTEBOOK round 6) -> the eight-wave kernel whose waves 4-7 fill while waves 0-3 multiply
    if (variant == GV_128 && auto_choice && 8L * (((M + GM - 1) / GM + 7) / 8) * (N / GN) <= (long)cu_count() && K / GK >= 4) variant = GV_PC;
    if ((ep & EP_PACK) && (!(ep & EP_PREAFF) || N % 3 || (N / 3) % 64 || M % 64 || aux.lt <= 0 || aux.lt % 64 || M % aux.lt || (aux.qst && aux.ldqst < M)))
        return fail(FOHO_ERR_BAD_ARG, "geo gemm: EP_PACK operands");
    if ((ep & EP_DELTA) && (!(ep & EP_TRANS) || !R || !aux.ndelta || aux.heads * 64 != N)) return fail(FOHO_ERR_BAD_ARG, "geo gemm: EP_DELTA operands");
    if (variant == GV_PHASED192 && (ep & EP_PACK) && aux.lt % 192) variant = GV_PHASED;   // (a 32-row part would straddle two images)
    const bool big = variant != GV_128 && variant != GV_DEEP && variant != GV_PC;
    const int tn = big ? HN : GN, tm = variant == GV_PHASED192 ? 192 : (big ? HM : GM);
    const int ntn = N / tn, ntm = (M + tm - 1) / tm;
    const dim3 grid(8 * ((ntm + 7) / 8) * ntn);
#define GEO_GEMM_CASE(E) case E: launch_gemm<E>(variant, grid, s, A, lda, Wt, ldw, bias, R, ldr, C, ldc, M, N, K, scale, C2, ldc2, Mdev, aux); break
    switch (ep) {
        GEO_GEMM_CASE(0);
        GEO_GEMM_CASE(EP_GELU);
        GEO_GEMM_CASE(EP_RESID);
        GEO_GEMM_CASE(EP_GELU | EP_SAVEZ);
        GEO_GEMM_CASE(EP_GELUBWD);
        GEO_GEMM_CASE(EP_TRANS);
        GEO_GEMM_CASE(EP_QNORM);
        GEO_GEMM_CASE(EP_TRANS | EP_QNORM);
        GEO_GEMM_CASE(EP_RESID | EP_STATS);
        GEO_GEMM_CASE(EP_GELU | EP_PREAFF);
        GEO_GEMM_CASE(EP_RESID | EP_LOGIT);
        // the ShapeVAE transformer's (foho_vae.inc): fused q | k | v behind a folded LayerNorm (+ qk_norm, + the un-normalised copy), fc1 keeping its pre-activation
        // (+ EP_PACK: the transposed / scaled copies the attention kernels stream; EP_TRANS | EP_DELTA: dO, dO^T and the backward's delta)
        GEO_GEMM_CASE(EP_PREAFF | EP_PACK);
        GEO_GEMM_CASE(EP_PREAFF | EP_QKN | EP_PACK);
        GEO_GEMM_CASE(EP_PREAFF | EP_QKN | EP_SAVEZ | EP_PACK);
        GEO_GEMM_CASE(EP_GELU | EP_PREAFF | EP_SAVEZ);
        GEO_GEMM_CASE(EP_TRANS | EP_DELTA);
        default: return fail(FOHO_ERR_BAD_ARG, "geo gemm: epilogue");
    }
#undef GEO_GEMM_CASE
    return launch_ok("k_geo_gemm") ? FOHO_OK : FOHO_ERR_LAUNCH;
}

static int check_weights(const foho_geo_weights* w) {
    if (!w) return fail(FOHO_ERR_BAD_ARG, "foho_geo: null weights");
    if (w->width <= 0 || w->width % 128 || w->width > 1024) return fail(FOHO_ERR_BAD_ARG, "foho_geo: width must be a multiple of 128, at most 1024");
    if (w->heads <= 0 || w->width != w->heads * 64) return fail(FOHO_ERR_BAD_ARG, "foho_geo: head dimension must be 64");
    if (w->n_latents <= 0 || w->n_latents % 64) return fail(FOHO_ERR_BAD_ARG, "foho_geo: n_latents must be a multiple of 64");
    if (w->hidden <= 0 || w->hidden % 128) return fail(FOHO_ERR_BAD_ARG, "foho_geo: hidden must be a multiple of 128");
    if (w->n_freqs < 0 || w->n_freqs > 10) return fail(FOHO_ERR_BAD_ARG, "foho_geo: 3 (2 n_freqs + 1) must fit 64 columns");
    if ((size_t)w->n_latents * w->width * 4 >= ((size_t)1 << 31)) return fail(FOHO_ERR_BAD_ARG, "foho_geo: K / V of the latent tokens exceed 32-bit buffer offsets");
    if (!w->w_qproj || !w->b_qproj || !w->w_q || !w->b_q || !w->w_kv || !w->b_kv || !w->w_proj || !w->b_proj || !w->w_fc1 || !w->b_fc1 ||
        !w->w_fc2 || !w->b_fc2 || !w->w_out || !w->ln_q_g || !w->ln_q_b || !w->ln_kv_g || !w->ln_kv_b || !w->ln_2_g || !w->ln_2_b ||
        !w->ln_post_g || !w->ln_post_b || !w->freqs)
        return fail(FOHO_ERR_BAD_ARG, "foho_geo: null weight pointer");
    return FOHO_OK;
}

struct Layout {
    size_t kv, vt, e, a, b, c, h, w1f, fold1, fold2, stats, rowstat, total;
};
static Layout layout(const foho_geo_weights* w, int chunk) {
    Layout l{};
    size_t off = 0;
    auto take = [&](size_t halfs) {
        const size_t o = off;
        off += (halfs * 2 + 255) & ~(size_t)255;
        return o;
    };
    const size_t W = w->width, Lr = w->n_latents, rows = std::max<size_t>(chunk, Lr);  // the prepare step normalises the latents in buffer A
    l.kv = take(Lr * 2 * W);
    l.vt = take(W * Lr);
    // LayerNorm folded into the forward GEMMs (chain_latent_side): fc1's weights with ln_2's gain and the folded vectors -- like kv / vt
    // written by the prepare step, at offsets that do not depend on the chunk
    l.w1f = take((size_t)w->hidden * W);
    l.fold1 = take((size_t)w->hidden * 4);        // 2 hidden floats
    l.fold2 = take((size_t)W * 4 + 8);            // 2 W + 2 floats
    l.e = take((size_t)chunk * 64);
    l.a = take(rows * W);
    l.b = take((size_t)chunk * W);
    l.c = take((size_t)chunk * W);
    l.h = take((size_t)chunk * w->hidden);
    // ... the per-row statistics of the folded LayerNorms (16 parts of 4 floats at width 1024) and (rstd, rstd mean) per row
    l.stats = take((size_t)chunk * (W / 64) * 8); // chunk x parts x 4 floats
    l.rowstat = take((size_t)chunk * 4);          // chunk x 2 floats
    l.total = off;
    return l;
}
struct Fold {   // device pointers of the folded operands in a prepared workspace (w1f == nullptr: chain with LayerNorm kernels)
    const h16* w1f = nullptr;
    const float *fold1 = nullptr, *fold2 = nullptr;
    float *stats = nullptr, *rowstat = nullptr;
};
static Fold fold_of(const foho_geo_weights* w, const Layout& l, char* base) {
    Fold f;
    if (w->flags & FOHO_GEO_NO_LNFUSE) return f;   // the forward chain with its LayerNorm kernels (A/B measurements, the parity test of the folded form)
    f.w1f = (const h16*)(base + l.w1f), f.fold1 = (const float*)(base + l.fold1), f.fold2 = (const float*)(base + l.fold2);
    f.stats = (float*)(base + l.stats), f.rowstat = (float*)(base + l.rowstat);
    return f;
}
static int fold_weights(const foho_geo_weights* w, const Layout& l, char* base, hipStream_t s) {
    const int W = w->width, F = w->hidden;
    hipLaunchKernelGGL(k_geo_fold_fc1, dim3((F + 3) / 4), dim3(256), 0, s, (const h16*)w->w_fc1, w->b_fc1, w->ln_2_g, w->ln_2_b, F, W, (h16*)(base + l.w1f),
                       (float*)(base + l.fold1));
    hipLaunchKernelGGL(k_geo_fold_post, dim3(1), dim3(256), 0, s, w->b_fc2, w->ln_post_g, w->ln_post_b, w->w_out, W, (float*)(base + l.fold2));
    return launch_ok("k_geo_fold") ? FOHO_OK : FOHO_ERR_LAUNCH;
}

}  // namespace geo

using namespace geo;

extern "C" const char* foho_geo_last_error(void) { return g_err; }

extern "C" int64_t foho_geo_abi_size(void) { return (int64_t)sizeof(foho_geo_weights); }

extern "C" size_t foho_geo_workspace_bytes(const foho_geo_weights* w, int32_t chunk_rows) {
    if (check_weights(w) != FOHO_OK || chunk_rows <= 0) return 0;
    return layout(w, chunk_rows).total;
}

extern "C" int foho_geo_prepare(const foho_geo_weights* w, const void* latents, int32_t chunk_rows, void* ws, size_t ws_bytes, void* stream_) {
    if (int rc = check_weights(w)) return rc;
    if (!latents || !ws || chunk_rows <= 0) return fail(FOHO_ERR_BAD_ARG, "foho_geo_prepare: null argument");
    const Layout l = layout(w, chunk_rows);
    if (ws_bytes < l.total) return fail(FOHO_ERR_WORKSPACE, "foho_geo_prepare: workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    char* base = (char*)ws;
    h16 *kv = (h16*)(base + l.kv), *vt = (h16*)(base + l.vt), *ln = (h16*)(base + l.a);
    const int W = w->width, Lr = w->n_latents;
    hipLaunchKernelGGL(k_geo_ln<0>, dim3((Lr + 3) / 4), dim3(256), 0, s, (const h16*)latents, W, w->ln_kv_g, w->ln_kv_b, ln, W, Lr, W, eps_of(w, w->ln_kv_eps),
                       (const float*)nullptr, 0.0f, (const float*)nullptr, 0.0f, 0.0f, 0.0f, (float*)nullptr, (const int*)nullptr);
    if (!launch_ok("k_geo_ln(kv)")) return FOHO_ERR_LAUNCH;
    if (int rc = gemm(0, ln, W, (const h16*)w->w_kv, W, w->b_kv, nullptr, 0, kv, 2 * W, Lr, 2 * W, W, 1.0f, s)) return rc;
    if (w->k_norm) hipLaunchKernelGGL(k_geo_knorm, dim3((Lr * w->heads + 255) / 256), dim3(256), 0, s, kv, 2 * W, Lr, w->heads, w->k_norm);
    hipLaunchKernelGGL(k_geo_pack_vt, dim3((W + 255) / 256, Lr / 16), dim3(256), 0, s, kv, 2 * W, W, Lr, vt);
    if (!launch_ok("k_geo_pack_vt")) return FOHO_ERR_LAUNCH;
    return fold_weights(w, l, base, s);
}

static void launch_attn(dim3 grid, hipStream_t s, const h16* Q, int ldq, const h16* Kp, int ldk, const h16* Vt, int L, h16* O, int ldo, int M, int heads, float* nlse,
                        const int* Mdev, int qhs = 64, int khs = 64, float qscale = 1.0f, float* lse_nat = nullptr, AttnB bs = AttnB{}) {
    hipLaunchKernelGGL(k_geo_attn, grid, dim3(256), 0, s, Q, ldq, Kp, ldk, Vt, L, O, ldo, M, heads, nlse, Mdev, qhs, khs, qscale, lse_nat, bs);
}

// ---- the forward chain in two halves: what depends only on the query points (Fourier embedding -> query projection -> ln_q ->
// c_q [+ q_norm], scaled for the exp2 softmax), and what depends on the latent tokens.  The first half is the same for every
// decode of one grid (foho_geo_prepare_queries caches it: X0 and Qs of all rows, 4 KB per query at width 1024).
static int chain_query_side(const foho_geo_weights* w, const float* q, int M, h16* E, h16* X0, h16* Xn, h16* Qs, h16* QsT, int ldt, hipStream_t s,
                            const int* Mdev = nullptr) {
    const int W = w->width;
    const float qscale = 1.4426950408889634f * 0.125f;  // log2(e) / sqrt(64): the attention kernel exponentiates with exp2
    const float* nof = nullptr;
    hipLaunchKernelGGL(k_geo_embed, dim3((M * 8 + 255) / 256), dim3(256), 0, s, q, M, w->n_freqs, w->freqs, E, Mdev);
    if (!launch_ok("k_geo_embed")) return FOHO_ERR_LAUNCH;
    if (int rc = gemm(0, E, 64, (const h16*)w->w_qproj, 64, w->b_qproj, nullptr, 0, X0, W, M, W, 64, 1.0f, s, nullptr, 0, Mdev)) return rc;
    hipLaunchKernelGGL(k_geo_ln<0>, dim3((M + 3) / 4), dim3(256), 0, s, X0, W, w->ln_q_g, w->ln_q_b, Xn, W, M, W, eps_of(w, w->ln_q_eps), nof, 0.0f, nof, 0.0f, 0.0f, 0.0f,
                       (float*)nullptr, Mdev);
    if (!launch_ok("k_geo_ln(q)")) return FOHO_ERR_LAUNCH;
    return gemm((QsT ? EP_TRANS : 0) | (w->q_norm ? EP_QNORM : 0), Xn, W, (const h16*)w->w_q, W, w->b_q, (const h16*)w->q_norm, 0, Qs, W, M, W, W, qscale, s, QsT,
                ldt, Mdev);
}

// attention over the latent tokens -> c_proj + residual -> ln_2 -> fc1 + GELU -> fc2 + residual.  Z != NULL keeps the MLP's
// pre-activation, lse != NULL the attention's log-sum-exp (both for a backward).  X0 / Qs: the query side's outputs for these rows.
//
// With `fold` (the plain forward: nothing kept, logits wanted) the two LayerNorms of this half never run as kernels:
//   ln_2 -> fc1:  LN(x) W^T = rstd (x (W gamma)^T) - rstd mean s + (b + W beta), s = row sums of W gamma -- c_proj's epilogue leaves the
//     row statistics of x1 (EP_STATS), fc1 multiplies the un-normalised x1 with the folded weights and applies (rstd, mean) per row in
//     its epilogue in front of the GELU (EP_PREAFF);
//   fc2 -> ln_post -> output_proj:  logit = rstd (x2 . gw - mean GW) + C0 + b_out, gw = gamma_post w_out -- fc2's epilogue reduces its
//     rows to (mean, M2, x2 . gw) per 64-column part (EP_LOGIT) and x2 is never written; k_geo_logit_finish merges the parts.
// Same fp16-rounded x1 / x2 as the chain with LayerNorm kernels; what differs is one rounding (LN output to fp16 there, W gamma to
// fp16 here) -- tests/test_geo_decode.py compares the two forms.  100 MB of reads + 100 MB of writes less per LayerNorm and row block.
static int chain_latent_side(const foho_geo_weights* w, int M, const h16* X0, const h16* Qs, h16* At, h16* X1, h16* Xn, h16* Z, h16* H, h16* X2, float* lse,
                             const h16* kv, const h16* vt, hipStream_t s, const int* Mdev = nullptr, const Fold* fold = nullptr, const float* q = nullptr,
                             float* logits = nullptr) {
    const int W = w->width, Lr = w->n_latents, F = w->hidden, NH = w->heads;
    const float* nof = nullptr;
    launch_attn(dim3(((M + AQ - 1) / AQ) * NH), s, Qs, W, kv, 2 * W, vt, Lr, At, W, M, NH, lse, Mdev);
    if (!launch_ok("k_geo_attn")) return FOHO_ERR_LAUNCH;
    if (fold && fold->w1f && !Z) {
        const int parts = W / 64;
        if (int rc = gemm(EP_RESID | EP_STATS, At, W, (const h16*)w->w_proj, W, w->b_proj, X0, W, X1, W, M, W, W, 1.0f, s, (h16*)fold->stats, parts, Mdev)) return rc;
        hipLaunchKernelGGL(k_geo_rowstat_finish, dim3((M + 15) / 16), dim3(256), 0, s, fold->stats, parts, M, eps_of(w, w->ln_2_eps), (float2*)fold->rowstat, Mdev);
        if (!launch_ok("k_geo_rowstat_finish")) return FOHO_ERR_LAUNCH;
        if (int rc = gemm(EP_GELU | EP_PREAFF, X1, W, fold->w1f, W, fold->fold1, (const h16*)fold->rowstat, F, H, F, M, F, W, 1.0f, s, nullptr, 0, Mdev)) return rc;
        if (int rc = gemm(EP_RESID | EP_LOGIT, H, F, (const h16*)w->w_fc2, F, fold->fold2, X1, W, X2, W, M, W, F, 1.0f, s, (h16*)fold->stats, parts, Mdev)) return rc;
        hipLaunchKernelGGL(k_geo_logit_finish, dim3((M + 15) / 16), dim3(256), 0, s, fold->stats, parts, M, w->ln_eps, fold->fold2 + 2 * W, w->b_out, q,
                           w->prior_radius, w->prior_sharpness, w->out_gain, logits, Mdev);
        return launch_ok("k_geo_logit_finish") ? FOHO_OK : FOHO_ERR_LAUNCH;
    }
    if (int rc = gemm(EP_RESID, At, W, (const h16*)w->w_proj, W, w->b_proj, X0, W, X1, W, M, W, W, 1.0f, s, nullptr, 0, Mdev)) return rc;
    hipLaunchKernelGGL(k_geo_ln<0>, dim3((M + 3) / 4), dim3(256), 0, s, X1, W, w->ln_2_g, w->ln_2_b, Xn, W, M, W, eps_of(w, w->ln_2_eps), nof, 0.0f, nof, 0.0f, 0.0f, 0.0f,
                       (float*)nullptr, Mdev);
    if (!launch_ok("k_geo_ln(2)")) return FOHO_ERR_LAUNCH;
    if (int rc = gemm(Z ? (EP_GELU | EP_SAVEZ) : EP_GELU, Xn, W, (const h16*)w->w_fc1, W, w->b_fc1, nullptr, 0, H, F, M, F, W, 1.0f, s, Z, Z ? F : 0, Mdev)) return rc;
    return gemm(EP_RESID, H, F, (const h16*)w->w_fc2, F, w->b_fc2, X1, W, X2, W, M, W, F, 1.0f, s, nullptr, 0, Mdev);
}

// logits = output_proj(ln_post(x2)) (+ the stand-in's analytic prior)
static int chain_logits(const foho_geo_weights* w, const float* q, int M, const h16* X2, float* logits, hipStream_t s) {
    hipLaunchKernelGGL(k_geo_ln<1>, dim3((M + 3) / 4), dim3(256), 0, s, X2, w->width, w->ln_post_g, w->ln_post_b, (h16*)nullptr, 0, M, w->width, w->ln_eps, w->w_out,
                       w->b_out, q, w->prior_radius, w->prior_sharpness, w->out_gain, logits, (const int*)nullptr);
    return launch_ok("k_geo_ln(post)") ? FOHO_OK : FOHO_ERR_LAUNCH;
}

extern "C" int foho_geo_decode_fwd(const foho_geo_weights* w, const float* queries, int64_t n_queries, float* logits, int32_t chunk_rows,
                                   void* ws, size_t ws_bytes, void* stream_) {
    if (int rc = check_weights(w)) return rc;
    if (!queries || !logits || !ws || chunk_rows <= 0 || n_queries < 0) return fail(FOHO_ERR_BAD_ARG, "foho_geo_decode_fwd: null argument");
    const Layout l = layout(w, chunk_rows);
    if (ws_bytes < l.total) return fail(FOHO_ERR_WORKSPACE, "foho_geo_decode_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    char* base = (char*)ws;
    const h16 *kv = (const h16*)(base + l.kv), *vt = (const h16*)(base + l.vt);
    h16 *E = (h16*)(base + l.e), *bA = (h16*)(base + l.a), *bB = (h16*)(base + l.b), *bC = (h16*)(base + l.c), *bH = (h16*)(base + l.h);
    const Fold fold = fold_of(w, l, base);
    for (int64_t r0 = 0; r0 < n_queries; r0 += chunk_rows) {
        const int M = (int)std::min<int64_t>(chunk_rows, n_queries - r0);
        const float* q = queries + 3 * r0;
        // x0 -> A, ln_q(x0) -> B, q -> C;  attention C -> B;  x1 = x0 + c_proj: B (+A) -> C;  ln_2: C -> B;  fc1: B -> H;  x2 = x1 + fc2: H (+C) -> A
        if (int rc = chain_query_side(w, q, M, E, bA, bB, bC, nullptr, 0, s)) return rc;
        if (int rc = chain_latent_side(w, M, bA, bC, bB, bC, bB, nullptr, bH, bA, nullptr, kv, vt, s, nullptr, &fold, q, logits + r0)) return rc;
        if (!fold.w1f)
            if (int rc = chain_logits(w, q, M, bA, logits + r0, s)) return rc;
    }
    return FOHO_OK;
}

// ---- the query side, cached (PL:1125-1143, 298-308: the grid of a guidance run never changes -- 65^3 points for all 550 inner
// iterations of every image): X0 = query_proj(embed(q)) and Qs = c_q(ln_q(X0)) [q_norm] log2(e) / 8 of ALL rows, fp16.
struct CacheLayout {
    size_t x0, qs, total;
};
static CacheLayout cache_layout(const foho_geo_weights* w, int64_t n) {
    CacheLayout l{};
    const size_t rows = ((size_t)std::max<int64_t>(n, 1) * w->width * 2 + 255) & ~(size_t)255;
    l.x0 = 0, l.qs = rows, l.total = 2 * rows;
    return l;
}
extern "C" size_t foho_geo_query_cache_bytes(const foho_geo_weights* w, int64_t n_queries) {
    if (check_weights(w) != FOHO_OK || n_queries < 0) return 0;
    return cache_layout(w, n_queries).total;
}

extern "C" int foho_geo_prepare_queries(const foho_geo_weights* w, const float* queries, int64_t n_queries, int32_t chunk_rows, void* ws, size_t ws_bytes,
                                        void* cache, size_t cache_bytes, void* stream_) {
    if (int rc = check_weights(w)) return rc;
    if (!queries || !ws || !cache || chunk_rows <= 0 || n_queries < 0) return fail(FOHO_ERR_BAD_ARG, "foho_geo_prepare_queries: null argument");
    const Layout l = layout(w, chunk_rows);
    const CacheLayout c = cache_layout(w, n_queries);
    if (ws_bytes < l.total) return fail(FOHO_ERR_WORKSPACE, "foho_geo_prepare_queries: workspace too small");
    if (cache_bytes < c.total) return fail(FOHO_ERR_WORKSPACE, "foho_geo_prepare_queries: cache too small (foho_geo_query_cache_bytes)");
    hipStream_t s = (hipStream_t)stream_;
    char* base = (char*)ws;
    h16 *E = (h16*)(base + l.e), *bB = (h16*)(base + l.b);
    h16 *X0 = (h16*)((char*)cache + c.x0), *Qs = (h16*)((char*)cache + c.qs);
    const size_t W = w->width;
    for (int64_t r0 = 0; r0 < n_queries; r0 += chunk_rows) {
        const int M = (int)std::min<int64_t>(chunk_rows, n_queries - r0);
        if (int rc = chain_query_side(w, queries + 3 * r0, M, E, X0 + r0 * W, bB, Qs + r0 * W, nullptr, 0, s)) return rc;
    }
    return FOHO_OK;
}

extern "C" int foho_geo_decode_fwd_cached(const foho_geo_weights* w, const float* queries, int64_t n_queries, const void* cache, size_t cache_bytes,
                                          float* logits, int32_t chunk_rows, void* ws, size_t ws_bytes, void* stream_) {
    if (int rc = check_weights(w)) return rc;
    if (!queries || !cache || !logits || !ws || chunk_rows <= 0 || n_queries < 0) return fail(FOHO_ERR_BAD_ARG, "foho_geo_decode_fwd_cached: null argument");
    const Layout l = layout(w, chunk_rows);
    const CacheLayout c = cache_layout(w, n_queries);
    if (ws_bytes < l.total) return fail(FOHO_ERR_WORKSPACE, "foho_geo_decode_fwd_cached: workspace too small");
    if (cache_bytes < c.total) return fail(FOHO_ERR_WORKSPACE, "foho_geo_decode_fwd_cached: cache too small");
    hipStream_t s = (hipStream_t)stream_;
    char* base = (char*)ws;
    const h16 *kv = (const h16*)(base + l.kv), *vt = (const h16*)(base + l.vt);
    h16 *bA = (h16*)(base + l.a), *bB = (h16*)(base + l.b), *bC = (h16*)(base + l.c), *bH = (h16*)(base + l.h);
    const Fold fold = fold_of(w, l, base);
    const h16 *X0 = (const h16*)((const char*)cache + c.x0), *Qs = (const h16*)((const char*)cache + c.qs);
    const size_t W = w->width;
    for (int64_t r0 = 0; r0 < n_queries; r0 += chunk_rows) {
        const int M = (int)std::min<int64_t>(chunk_rows, n_queries - r0);
        // attention Qs -> B;  x1 = x0 + c_proj: B (+X0) -> C;  ln_2: C -> B;  fc1: B -> H;  x2 = x1 + fc2: H (+C) -> A
        if (int rc = chain_latent_side(w, M, X0 + r0 * W, Qs + r0 * W, bB, bC, bB, nullptr, bH, bA, nullptr, kv, vt, s, nullptr, &fold, queries + 3 * r0, logits + r0))
            return rc;
        if (!fold.w1f)
            if (int rc = chain_logits(w, queries + 3 * r0, M, bA, logits + r0, s)) return rc;
    }
    return FOHO_OK;
}

struct BwdLayout {
    size_t e, x0, xn, qs, qst, at, x1, z, h, x2, dx2, dat, lse, delta, part, total;
    int ldt, splits;
};
// workgroups of k_geo_attn_bwd per (key block, head): enough for ~4 per CU, preferring a count that fills whole rounds of the
// 512 resident workgroups (2 per CU)
static int bwd_splits(const foho_geo_weights* w, int chunk) {
    const int ntiles = (chunk + 63) / 64, base = (w->n_latents / 128) * w->heads;
    int s0 = std::max(1, (1024 + base - 1) / base), best = s0;
    double bw = 1e9;
    for (int sp = s0; sp < s0 + 4; sp++) {
        const double n = (double)base * sp, waste = std::ceil(n / 512.0) * 512.0 / n;
        if (waste < bw - 1e-9) bw = waste, best = sp;
    }
    return std::max(1, std::min(best, ntiles));
}
static BwdLayout bwd_layout(const foho_geo_weights* w, int chunk) {
    BwdLayout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t W = w->width, F = w->hidden, C = chunk;
    l.ldt = (chunk + 63) & ~63;
    l.e = take(C * 64 * 2);
    l.x0 = take(C * W * 2);
    l.xn = take(C * W * 2);
    l.qs = take(C * W * 2);
    l.qst = take(W * (size_t)l.ldt * 2);
    l.at = take(C * W * 2);
    l.x1 = take(C * W * 2);
    l.z = take(C * F * 2);
    l.h = take(C * F * 2);
    l.x2 = take(C * W * 2);
    l.dx2 = take(C * W * 2);
    l.dat = take(W * (size_t)l.ldt * 2);
    l.lse = take((size_t)l.ldt * w->heads * 4);    // rows up to the next multiple of 64: the padding of the last query tile
    l.delta = take((size_t)l.ldt * w->heads * 4);
    l.splits = bwd_splits(w, chunk);
    l.part = take((size_t)l.splits * w->n_latents * 2 * W * 4);
    l.total = off;
    return l;
}

extern "C" size_t foho_geo_bwd_workspace_bytes(const foho_geo_weights* w, int32_t chunk_rows) {
    if (check_weights(w) != FOHO_OK || chunk_rows <= 0) return 0;
    return bwd_layout(w, chunk_rows).total;
}

extern "C" int foho_geo_set_kv(const foho_geo_weights* w, const void* kv_in, int32_t chunk_rows, void* ws, size_t ws_bytes, void* stream_) {
    if (int rc = check_weights(w)) return rc;
    if (!kv_in || !ws || chunk_rows <= 0) return fail(FOHO_ERR_BAD_ARG, "foho_geo_set_kv: null argument");
    const Layout l = layout(w, chunk_rows);
    if (ws_bytes < l.total) return fail(FOHO_ERR_WORKSPACE, "foho_geo_set_kv: workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    char* base = (char*)ws;
    h16 *kv = (h16*)(base + l.kv), *vt = (h16*)(base + l.vt);
    const int W = w->width, Lr = w->n_latents;
    if (hipMemcpyAsync(kv, kv_in, (size_t)Lr * 2 * W * 2, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(FOHO_ERR_LAUNCH, "foho_geo_set_kv: copy failed");
    hipLaunchKernelGGL(k_geo_pack_vt, dim3((W + 255) / 256, Lr / 16), dim3(256), 0, s, kv, 2 * W, W, Lr, vt);
    if (!launch_ok("k_geo_pack_vt")) return FOHO_ERR_LAUNCH;
    return fold_weights(w, l, base, s);
}

// What the backward of one row block needs from its forward: per block in the `save` buffer (kept mode) or in the backward
// workspace (recompute mode)
struct SavedLayout {
    size_t qs, qst, at, x1, z, x2, lse, total;
};
static SavedLayout saved_layout(const foho_geo_weights* w, int chunk) {
    SavedLayout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t W = w->width, F = w->hidden, C = chunk, ldt = (chunk + 63) & ~63;
    l.qs = take(C * W * 2);
    l.qst = take(W * ldt * 2);
    l.at = take(C * W * 2);
    l.x1 = take(C * W * 2);
    l.z = take(C * F * 2);
    l.x2 = take(C * W * 2);
    l.lse = take(ldt * (size_t)w->heads * 4);
    l.total = off;
    return l;
}
struct ChunkPtrs {
    h16 *E, *X0, *Xn, *H, *dX2, *dAT;        // scratch
    h16 *Qs, *QsT, *At, *X1, *Z, *X2;        // saved by the forward
    float *lse, *delta;
    int ldt;
};

// the forward chain of one row block with everything its backward needs kept (P.Qs .. P.lse)
static int chain_fwd_keep(const foho_geo_weights* w, const float* q, int M, const ChunkPtrs& P, const h16* kv, const h16* vt, hipStream_t s,
                          const int* Mdev = nullptr) {
    if (int rc = chain_query_side(w, q, M, P.E, P.X0, P.Xn, P.Qs, P.QsT, P.ldt, s, Mdev)) return rc;
    return chain_latent_side(w, M, P.X0, P.Qs, P.At, P.X1, P.Xn, P.Z, P.H, P.X2, P.lse, kv, vt, s, Mdev);
}

// ... and backwards: logits -> ln_post -> fc2 -> GELU -> fc1 -> ln_2 (+ residual) -> c_proj -> attention (K, V partial sums)
static int chain_bwd(const foho_geo_weights* w, const float* grad_logits, int M, const ChunkPtrs& P, const h16* kv, int splits, int accumulate, float* part,
                     hipStream_t s, const int* Mdev = nullptr) {
    const int W = w->width, Lr = w->n_latents, F = w->hidden, NH = w->heads;
    const float* nof = nullptr;
    const dim3 rows((M + 3) / 4), blk(256);
    hipLaunchKernelGGL(k_geo_ln_bwd<1>, rows, blk, 0, s, P.X2, w->ln_post_g, (const h16*)nullptr, (const h16*)nullptr, grad_logits, w->out_gain, w->w_out, P.dX2,
                       M, W, w->ln_eps, Mdev);
    if (int rc = gemm(EP_GELUBWD, P.dX2, W, (const h16*)w->w_fc2_t, W, w->zeros, P.Z, F, P.H, F, M, F, W, 1.0f, s, nullptr, 0, Mdev)) return rc;          // dZ -> H
    if (int rc = gemm(0, P.H, F, (const h16*)w->w_fc1_t, F, w->zeros, nullptr, 0, P.Xn, W, M, W, F, 1.0f, s, nullptr, 0, Mdev)) return rc;                 // d ln_2 out -> Xn
    hipLaunchKernelGGL(k_geo_ln_bwd<0>, rows, blk, 0, s, P.X1, w->ln_2_g, P.Xn, P.dX2, nof, 0.0f, nof, P.X0, M, W, eps_of(w, w->ln_2_eps), Mdev);                    // dX1 -> X0
    if (int rc = gemm(EP_TRANS, P.X0, W, (const h16*)w->w_proj_t, W, w->zeros, nullptr, 0, P.dX2, W, M, W, W, 1.0f, s, P.dAT, P.ldt, Mdev)) return rc;    // dO -> dX2
    hipLaunchKernelGGL(k_geo_delta, dim3((((M + 63) & ~63) + 3) / 4), blk, 0, s, P.dX2, P.At, W, NH, M, P.delta, Mdev);
    if (!launch_ok("geometry decoder backward chain (row kernels)")) return FOHO_ERR_LAUNCH;
    const int nkb = Lr / 128;
    hipLaunchKernelGGL(k_geo_attn_bwd, dim3(8 * ((NH * splits + 7) / 8) * nkb), blk, 0, s, P.Qs, P.QsT, P.dX2, P.dAT, P.ldt, P.lse, P.delta, kv, 2 * W, W, NH, M, splits, Lr,
                       accumulate, part, Mdev);
    return launch_ok("k_geo_attn_bwd") ? FOHO_OK : FOHO_ERR_LAUNCH;
}

static ChunkPtrs chunk_ptrs(const BwdLayout& b, char* base, const SavedLayout& sl, char* saved) {
    ChunkPtrs P;
    P.E = (h16*)(base + b.e), P.X0 = (h16*)(base + b.x0), P.Xn = (h16*)(base + b.xn), P.H = (h16*)(base + b.h), P.dX2 = (h16*)(base + b.dx2);
    P.dAT = (h16*)(base + b.dat), P.delta = (float*)(base + b.delta), P.ldt = b.ldt;
    if (saved) {
        P.Qs = (h16*)(saved + sl.qs), P.QsT = (h16*)(saved + sl.qst), P.At = (h16*)(saved + sl.at), P.X1 = (h16*)(saved + sl.x1);
        P.Z = (h16*)(saved + sl.z), P.X2 = (h16*)(saved + sl.x2), P.lse = (float*)(saved + sl.lse);
    } else {
        P.Qs = (h16*)(base + b.qs), P.QsT = (h16*)(base + b.qst), P.At = (h16*)(base + b.at), P.X1 = (h16*)(base + b.x1);
        P.Z = (h16*)(base + b.z), P.X2 = (h16*)(base + b.x2), P.lse = (float*)(base + b.lse);
    }
    return P;
}

extern "C" size_t foho_geo_saved_bytes(const foho_geo_weights* w, int32_t chunk_rows, int64_t n_queries) {
    if (check_weights(w) != FOHO_OK || chunk_rows <= 0 || n_queries < 0) return 0;
    const int64_t nchunks = (n_queries + chunk_rows - 1) / chunk_rows;
    return (size_t)std::max<int64_t>(nchunks, 1) * saved_layout(w, chunk_rows).total;
}

static int bwd_args(const char* who, const foho_geo_weights* w, const void* a, const void* b_, const void* ws, const void* bws, int32_t chunk_rows,
                    int64_t n_queries, bool need_t) {
    if (int rc = check_weights(w)) return rc;
    if (!a || !b_ || !ws || !bws || chunk_rows <= 0 || n_queries < 0) return fail(FOHO_ERR_BAD_ARG, std::string(who) + ": null argument");
    if (need_t && (!w->w_fc2_t || !w->w_fc1_t || !w->w_proj_t || !w->zeros)) return fail(FOHO_ERR_BAD_ARG, std::string(who) + ": transposed weights / zeros missing");
    if (w->n_latents % 128) return fail(FOHO_ERR_BAD_ARG, std::string(who) + ": n_latents must be a multiple of 128");
    return FOHO_OK;
}

extern "C" int foho_geo_decode_fwd_keep(const foho_geo_weights* w, const float* queries, int64_t n_queries, float* logits, int32_t chunk_rows, void* ws,
                                        size_t ws_bytes, void* bws, size_t bws_bytes, void* saved, size_t saved_bytes, void* stream_) {
    if (int rc = bwd_args("foho_geo_decode_fwd_keep", w, queries, logits, ws, bws, chunk_rows, n_queries, false)) return rc;
    const Layout l = layout(w, chunk_rows);
    const BwdLayout b = bwd_layout(w, chunk_rows);
    const SavedLayout sl = saved_layout(w, chunk_rows);
    if (ws_bytes < l.total || bws_bytes < b.total) return fail(FOHO_ERR_WORKSPACE, "foho_geo_decode_fwd_keep: workspace too small");
    if (!saved || saved_bytes < foho_geo_saved_bytes(w, chunk_rows, n_queries)) return fail(FOHO_ERR_WORKSPACE, "foho_geo_decode_fwd_keep: buffer for the activations too small");
    hipStream_t s = (hipStream_t)stream_;
    const h16 *kv = (const h16*)((char*)ws + l.kv), *vt = (const h16*)((char*)ws + l.vt);
    for (int64_t r0 = 0, c = 0; r0 < n_queries; r0 += chunk_rows, c++) {
        const int M = (int)std::min<int64_t>(chunk_rows, n_queries - r0);
        const ChunkPtrs P = chunk_ptrs(b, (char*)bws, sl, (char*)saved + c * sl.total);
        // columns of Qs^T beyond a ragged block's rows meet P = 0 in the backward and must be finite
        if ((M & 63) && hipMemsetAsync(P.QsT, 0, (size_t)w->width * b.ldt * 2, s) != hipSuccess) return fail(FOHO_ERR_LAUNCH, "foho_geo_decode_fwd_keep: memset failed");
        if (int rc = chain_fwd_keep(w, queries + 3 * r0, M, P, kv, vt, s)) return rc;
        if (int rc = chain_logits(w, queries + 3 * r0, M, P.X2, logits + r0, s)) return rc;
    }
    return FOHO_OK;
}

extern "C" int foho_geo_decode_bwd(const foho_geo_weights* w, const float* queries, int64_t n_queries, const float* grad_logits, float* grad_kv,
                                   int32_t chunk_rows, void* ws, size_t ws_bytes, void* bws, size_t bws_bytes, const void* saved, size_t saved_bytes,
                                   void* stream_) {
    if (int rc = bwd_args("foho_geo_decode_bwd", w, queries, grad_logits, ws, bws, chunk_rows, n_queries, true)) return rc;
    if (!grad_kv) return fail(FOHO_ERR_BAD_ARG, "foho_geo_decode_bwd: null argument");
    const Layout l = layout(w, chunk_rows);
    const BwdLayout b = bwd_layout(w, chunk_rows);
    const SavedLayout sl = saved_layout(w, chunk_rows);
    if (ws_bytes < l.total || bws_bytes < b.total) return fail(FOHO_ERR_WORKSPACE, "foho_geo_decode_bwd: workspace too small");
    if (saved && saved_bytes < foho_geo_saved_bytes(w, chunk_rows, n_queries)) return fail(FOHO_ERR_WORKSPACE, "foho_geo_decode_bwd: buffer of activations too small");
    hipStream_t s = (hipStream_t)stream_;
    const h16 *kv = (const h16*)((char*)ws + l.kv), *vt = (const h16*)((char*)ws + l.vt);
    char* base = (char*)bws;
    const int W = w->width, Lr = w->n_latents;
    // columns of the transposed copies beyond a ragged block's rows must be finite (they meet P = 0): clear them once
    if (hipMemsetAsync(base + b.dat, 0, (size_t)W * b.ldt * 2, s) != hipSuccess || (!saved && hipMemsetAsync(base + b.qst, 0, (size_t)W * b.ldt * 2, s) != hipSuccess))
        return fail(FOHO_ERR_LAUNCH, "foho_geo_decode_bwd: memset failed");
    float* part = (float*)(base + b.part);
    if (n_queries == 0) return hipMemsetAsync(grad_kv, 0, (size_t)Lr * 2 * W * 4, s) == hipSuccess ? FOHO_OK : fail(FOHO_ERR_LAUNCH, "foho_geo_decode_bwd: memset failed");
    for (int64_t r0 = 0, c = 0; r0 < n_queries; r0 += chunk_rows, c++) {
        const int M = (int)std::min<int64_t>(chunk_rows, n_queries - r0);
        const ChunkPtrs P = chunk_ptrs(b, base, sl, saved ? (char*)const_cast<void*>(saved) + c * sl.total : nullptr);
        if (!saved)
            if (int rc = chain_fwd_keep(w, queries + 3 * r0, M, P, kv, vt, s)) return rc;
        if (int rc = chain_bwd(w, grad_logits + r0, M, P, kv, b.splits, r0 > 0 ? 1 : 0, part, s)) return rc;
    }
    const size_t n4 = (size_t)Lr * 2 * W / 4;
    hipLaunchKernelGGL(k_geo_dkv_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, b.splits, n4, grad_kv);
    return launch_ok("k_geo_dkv_reduce") ? FOHO_OK : FOHO_ERR_LAUNCH;
}

// ---- backward over the ACTIVE rows only.  The gradient that reaches latent2sdf comes out of FlexiCubes (PL:1507-1509, 1600): it is
// non-zero on the end points of the grid edges the iso-surface crosses -- 5-10 % of the 65^3 rows.  A row whose logit gradient is zero
// contributes exactly zero to dK / dV, so the rows with grad_logits != 0 are compacted ON THE DEVICE (in row order: deterministic),
// the forward chain is recomputed for them and the backward runs on them.  The count stays in device memory: every launch is sized
// for a row block of the capacity and works on min(block, what is left of the count) rows -- blocks beyond the count leave at once --
// so there is no host synchronisation and the call can be captured in a hipGraph.
struct RowsLayout {
    size_t counts, mblk, q, g, idx, total;
    int nscan, nblk;
};
constexpr int ROWS_PER_WG = 2048;
static RowsLayout rows_layout(int64_t n, int64_t cap, int chunk) {
    RowsLayout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    l.nscan = (int)((n + ROWS_PER_WG - 1) / ROWS_PER_WG);
    l.nblk = (int)std::max<int64_t>((cap + chunk - 1) / chunk, 1);
    l.counts = take((size_t)std::max(l.nscan, 1) * 4);
    l.mblk = take((size_t)(std::max(l.nblk, 1) + 2) * 4);     // [0]: the count, [1]: rows dropped for lack of capacity, [2 + c]: rows of block c
    l.q = take((size_t)std::max<int64_t>(cap, 1) * 12);
    l.g = take((size_t)std::max<int64_t>(cap, 1) * 4);
    l.idx = take((size_t)std::max<int64_t>(cap, 1) * 4);
    l.total = off;
    return l;
}

extern "C" size_t foho_geo_rows_workspace_bytes(int64_t n_queries, int64_t row_cap, int32_t chunk_rows) {
    if (n_queries < 0 || chunk_rows <= 0) return 0;
    if (row_cap <= 0 || row_cap > n_queries) row_cap = n_queries;
    return rows_layout(n_queries, row_cap, chunk_rows).total;
}

extern "C" int foho_geo_decode_bwd_rows(const foho_geo_weights* w, const float* queries, int64_t n_queries, const float* grad_logits, float* grad_kv,
                                        int64_t row_cap, int32_t chunk_rows, void* ws, size_t ws_bytes, void* bws, size_t bws_bytes, void* rows_ws,
                                        size_t rows_ws_bytes, int32_t* stats_out, void* stream_) {
    if (int rc = bwd_args("foho_geo_decode_bwd_rows", w, queries, grad_logits, ws, bws, chunk_rows, n_queries, true)) return rc;
    if (!grad_kv || !rows_ws) return fail(FOHO_ERR_BAD_ARG, "foho_geo_decode_bwd_rows: null argument");
    if (n_queries >= ((int64_t)1 << 31)) return fail(FOHO_ERR_BAD_ARG, "foho_geo_decode_bwd_rows: row indices are 32-bit");
    if (row_cap <= 0 || row_cap > n_queries) row_cap = n_queries;
    const Layout l = layout(w, chunk_rows);
    const BwdLayout b = bwd_layout(w, chunk_rows);
    const SavedLayout sl = saved_layout(w, chunk_rows);
    const RowsLayout rl = rows_layout(n_queries, row_cap, chunk_rows);
    if (ws_bytes < l.total || bws_bytes < b.total || rows_ws_bytes < rl.total) return fail(FOHO_ERR_WORKSPACE, "foho_geo_decode_bwd_rows: workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    const h16 *kv = (const h16*)((char*)ws + l.kv), *vt = (const h16*)((char*)ws + l.vt);
    char *base = (char*)bws, *rb = (char*)rows_ws;
    const int W = w->width, Lr = w->n_latents;
    int *counts = (int*)(rb + rl.counts), *mblk = (int*)(rb + rl.mblk), *idx = (int*)(rb + rl.idx);
    float *qa = (float*)(rb + rl.q), *ga = (float*)(rb + rl.g);
    // the transposed copies' columns beyond a ragged block's rows must be finite (they meet P = 0): cleared by a kernel (memset nodes
    // of a captured graph are not reliably ordered on this stack)
    const size_t n16 = (size_t)W * b.ldt * 2 / 16;
    hipLaunchKernelGGL(k_geo_zero16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (uint4*)(base + b.dat), n16);
    hipLaunchKernelGGL(k_geo_zero16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (uint4*)(base + b.qst), n16);
    hipLaunchKernelGGL(k_geo_rows_count, dim3(std::max(rl.nscan, 1)), dim3(256), 0, s, grad_logits, n_queries, counts);
    hipLaunchKernelGGL(k_geo_rows_compact, dim3(std::max(rl.nscan, 1)), dim3(256), 0, s, grad_logits, queries, n_queries, counts, rl.nscan, row_cap, chunk_rows, rl.nblk,
                       mblk, qa, ga, idx, stats_out);
    if (!launch_ok("k_geo_rows_compact")) return FOHO_ERR_LAUNCH;
    float* part = (float*)(base + b.part);
    const ChunkPtrs P = chunk_ptrs(b, base, sl, nullptr);
    for (int64_t r0 = 0, c = 0; c < std::max(rl.nblk, 1); r0 += chunk_rows, c++) {
        const int M = (int)std::max<int64_t>(std::min<int64_t>(chunk_rows, row_cap - r0), 1);
        const int* Mdev = mblk + 2 + c;
        if (int rc = chain_fwd_keep(w, qa + 3 * r0, M, P, kv, vt, s, Mdev)) return rc;
        if (int rc = chain_bwd(w, ga + r0, M, P, kv, b.splits, c > 0 ? 1 : 0, part, s, Mdev)) return rc;
    }
    const size_t n4 = (size_t)Lr * 2 * W / 4;
    hipLaunchKernelGGL(k_geo_dkv_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, b.splits, n4, grad_kv);
    return launch_ok("k_geo_dkv_reduce") ? FOHO_OK : FOHO_ERR_LAUNCH;
}

// ---- Attention as an operator of its own, forward and backward (round 5): O = softmax(Q K^T / 8) V per head of 64, what
// torch.nn.functional.scaled_dot_product_attention computes for the self-attention layers of the ShapeVAE transformer that latent2sdf
// runs in front of the decoder (PL:295; sixteen layers of 3072 tokens x 16 heads, forward and backward in every inner iteration).
// The forward is the decoder's k_geo_attn, dK / dV its k_geo_attn_bwd, dQ k_geo_attn_dq.
struct SdpaLayout {
    size_t vt, kt, qs, qst, dot, delta, part, total;
    int ldt, splits;
};
static SdpaLayout sdpa_layout(int M, int L, int heads) {
    SdpaLayout l{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const size_t W = (size_t)heads * 64;
    l.ldt = (M + 63) & ~63;
    l.vt = take(W * L * 2);
    l.kt = take(W * L * 2);
    l.qs = take(W * (size_t)M * 2);
    l.qst = take(W * (size_t)l.ldt * 2);
    l.dot = take(W * (size_t)l.ldt * 2);
    l.delta = take((size_t)l.ldt * heads * 4);
    const int ntiles = (M + 63) / 64, base = std::max(1, (L / 128) * heads);
    l.splits = std::max(1, std::min((1024 + base - 1) / base, ntiles));
    l.part = take((size_t)l.splits * L * 2 * W * 4);
    l.total = off;
    return l;
}
static int sdpa_args(const char* who, const foho_sdpa_desc* d, bool bwd) {
    if (!d) return fail(FOHO_ERR_BAD_ARG, std::string(who) + ": null descriptor");
    const int M = d->M, L = d->L, heads = d->heads;
    if (M < 1 || L < 64 || L % 64 || heads < 1 || heads > 16 || d->batch < 1)
        return fail(FOHO_ERR_BAD_ARG, std::string(who) + ": M >= 1, L a multiple of 64, 1..16 heads of 64, batch >= 1");
    if (bwd && L % 128) return fail(FOHO_ERR_BAD_ARG, std::string(who) + ": L must be a multiple of 128");
    if (d->q_row < 64 || d->kv_row < 64 || d->q_head < 64 || d->kv_head < 64 || (d->q_row & 7) || (d->kv_row & 7) || (d->q_head & 7) || (d->kv_head & 7) ||
        (d->q_batch & 7) || (d->kv_batch & 7))
        return fail(FOHO_ERR_BAD_ARG, std::string(who) + ": strides must be multiples of 8 halfs (16-byte rows), at least 64");
    if (((size_t)(L - 1) * d->kv_row + (size_t)(heads - 1) * d->kv_head + 64) * 2 >= ((size_t)1 << 31) ||
        ((size_t)(M - 1) * d->q_row + (size_t)(heads - 1) * d->q_head + 64) * 2 >= ((size_t)1 << 31))
        return fail(FOHO_ERR_BAD_ARG, std::string(who) + ": operand too large for 32-bit buffer offsets");
    return FOHO_OK;
}
static const float SDPA_QSCALE = 1.4426950408889634f * 0.125f;   // log2(e) / sqrt(64): the kernels exponentiate with exp2

extern "C" size_t foho_sdpa_workspace_bytes(int32_t M, int32_t L, int32_t heads) {
    foho_sdpa_desc d{};
    d.M = M, d.L = L, d.heads = heads, d.batch = 1, d.q_row = d.kv_row = heads * 64, d.q_head = d.kv_head = 64;
    if (sdpa_args("foho_sdpa_workspace_bytes", &d, false) != FOHO_OK) return 0;
    return sdpa_layout(M, L, heads).total;
}
extern "C" int foho_sdpa_fwd(const foho_sdpa_desc* d, const void* q, const void* k, const void* v, void* out, float* nlse, float* lse_nat, void* ws,
                             size_t ws_bytes, void* stream_) {
    if (int rc = sdpa_args("foho_sdpa_fwd", d, false)) return rc;
    if (!q || !k || !v || !out || !ws) return fail(FOHO_ERR_BAD_ARG, "foho_sdpa_fwd: null argument");
    const int M = d->M, L = d->L, heads = d->heads, W = heads * 64;
    const SdpaLayout l = sdpa_layout(M, L, heads);
    if (ws_bytes < l.total) return fail(FOHO_ERR_WORKSPACE, "foho_sdpa_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    h16* vt = (h16*)((char*)ws + l.vt);
    const size_t nl = (size_t)((M + 63) & ~63) * heads;
    for (int b = 0; b < d->batch; b++) {
        const h16 *qb = (const h16*)q + (size_t)b * d->q_batch, *kb = (const h16*)k + (size_t)b * d->kv_batch, *vb = (const h16*)v + (size_t)b * d->kv_batch;
        hipLaunchKernelGGL(k_geo_pack_vt, dim3((W + 255) / 256, L / 16), dim3(256), 0, s, vb, (int)d->kv_row, W, L, vt, 0, (int)d->kv_head);
        launch_attn(dim3(((M + AQ - 1) / AQ) * heads), s, qb, (int)d->q_row, kb, (int)d->kv_row, (const h16*)vt, L, (h16*)out + (size_t)b * M * W, W, M, heads,
                    nlse ? nlse + b * nl : nullptr, (const int*)nullptr, (int)d->q_head, (int)d->kv_head, SDPA_QSCALE, lse_nat ? lse_nat + (size_t)b * heads * M : nullptr);
    }
    return launch_ok("k_geo_attn") ? FOHO_OK : FOHO_ERR_LAUNCH;
}
// The backward of one image: dQ, dK, dV from q / k / v where they lie, the forward's output and -lse, and dO (M, 64 heads; contiguous rows).
// gq rows have stride ldgq, gk / gv rows ldgkv (heads side by side in all three).
static int sdpa_bwd_image(const SdpaLayout& l, char* base, const h16* qb, int q_row, int q_head, const h16* kb, const h16* vb, int kvr, int kvh, const h16* O,
                          const h16* dO, const float* nls, h16* gq, int ldgq, h16* gk, h16* gv, int ldgkv, int M, int L, int heads, hipStream_t s) {
    const int W = heads * 64, nkb = L / 128;
    h16 *kt = (h16*)(base + l.kt), *qs = (h16*)(base + l.qs), *qst = (h16*)(base + l.qst), *dot = (h16*)(base + l.dot);
    float *delta = (float*)(base + l.delta), *part = (float*)(base + l.part);
    hipLaunchKernelGGL(k_geo_pack_vt, dim3((W + 255) / 256, L / 16), dim3(256), 0, s, kb, kvr, W, L, kt, 0, kvh);      // K^T, key-permuted
    if (M & 63) {   // columns of the transposed copies beyond M meet P = 0 and must be finite
        const size_t n16 = (size_t)W * l.ldt * 2 / 16;
        hipLaunchKernelGGL(k_geo_zero16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (uint4*)qst, n16);
        hipLaunchKernelGGL(k_geo_zero16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (uint4*)dot, n16);
    }
    const unsigned tb = (unsigned)(((size_t)M * (W / 8) + 255) / 256);
    hipLaunchKernelGGL(k_geo_transpose_perm, dim3(tb), dim3(256), 0, s, qb, q_row, M, W, qst, l.ldt, q_head, SDPA_QSCALE, qs);
    hipLaunchKernelGGL(k_geo_transpose_perm, dim3(tb), dim3(256), 0, s, dO, W, M, W, dot, l.ldt);
    hipLaunchKernelGGL(k_geo_delta, dim3((((M + 63) & ~63) + 3) / 4), dim3(256), 0, s, dO, O, W, heads, M, delta, (const int*)nullptr);
    if (!launch_ok("sdpa backward (row kernels)")) return FOHO_ERR_LAUNCH;
    // enough (head, key block) workgroups to fill the chip on their own: each walks ALL query tiles and writes its dK / dV itself (no
    // partial sums, no reduction pass: -14 us at 16 heads x 3072 keys); otherwise splits of the query tiles + k_geo_dkv_reduce16
    const bool direct = (long)heads * nkb >= 256;
    const int splits = direct ? 1 : l.splits;
    hipLaunchKernelGGL(k_geo_attn_bwd, dim3(8 * ((heads * splits + 7) / 8) * nkb), dim3(256), 0, s, (const h16*)qs, (const h16*)qst, dO, (const h16*)dot, l.ldt, nls,
                       (const float*)delta, kb, kvr, W, heads, M, splits, L, 0, part, (const int*)nullptr, vb, kvh, direct ? gk : (h16*)nullptr, direct ? gv : (h16*)nullptr, ldgkv);
    if (!direct) {
        const size_t n4 = (size_t)L * 2 * W / 4;
        hipLaunchKernelGGL(k_geo_dkv_reduce16, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, splits, L, W, gk, gv, ldgkv);
    }
    hipLaunchKernelGGL(k_geo_attn_dq, dim3(((M + DQQ - 1) / DQQ) * heads), dim3(256), 0, s, (const h16*)qs, dO, W, kb, kvr, W, (const h16*)kt, L, nls, (const float*)delta,
                       gq, M, heads, vb, kvh, ldgq);
    return launch_ok("k_geo_attn_dq") ? FOHO_OK : FOHO_ERR_LAUNCH;
}
extern "C" int foho_sdpa_bwd(const foho_sdpa_desc* d, const void* q, const void* k, const void* v, const void* out, const float* nlse, const void* grad_out,
                             void* grad_q, void* grad_k, void* grad_v, void* ws, size_t ws_bytes, void* stream_) {
    if (int rc = sdpa_args("foho_sdpa_bwd", d, true)) return rc;
    if (!q || !k || !v || !out || !nlse || !grad_out || !grad_q || !grad_k || !grad_v || !ws) return fail(FOHO_ERR_BAD_ARG, "foho_sdpa_bwd: null argument");
    const int M = d->M, L = d->L, heads = d->heads, W = heads * 64;
    const SdpaLayout l = sdpa_layout(M, L, heads);
    if (ws_bytes < l.total) return fail(FOHO_ERR_WORKSPACE, "foho_sdpa_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream_;
    const size_t nl = (size_t)((M + 63) & ~63) * heads;
    for (int b = 0; b < d->batch; b++) {
        const h16 *qb = (const h16*)q + (size_t)b * d->q_batch, *kb = (const h16*)k + (size_t)b * d->kv_batch, *vb = (const h16*)v + (size_t)b * d->kv_batch;
        if (int rc = sdpa_bwd_image(l, (char*)ws, qb, (int)d->q_row, (int)d->q_head, kb, vb, (int)d->kv_row, (int)d->kv_head, (const h16*)out + (size_t)b * M * W,
                                    (const h16*)grad_out + (size_t)b * M * W, nlse + b * nl, (h16*)grad_q + (size_t)b * M * W, W, (h16*)grad_k + (size_t)b * L * W,
                                    (h16*)grad_v + (size_t)b * L * W, W, M, L, heads, s))
            return rc;
    }
    return FOHO_OK;
}

// Unit entry points (tests / profiling): the GEMM and the attention kernel on their own.
extern "C" int foho_geo_gemm(const void* A, const void* Wt, const float* bias, const void* R, void* C, int32_t M, int32_t N, int32_t K,
                             int32_t gelu, float scale, void* stream) {
    if (!A || !Wt || !bias || !C) return fail(FOHO_ERR_BAD_ARG, "foho_geo_gemm: null argument");
    if ((gelu & 1) && R) return fail(FOHO_ERR_BAD_ARG, "foho_geo_gemm: GELU and a residual are not combined on this path");
    const int variant = (gelu & 2) ? GV_128 : (gelu & 4) ? GV_LOCKSTEP : (gelu & 8) ? GV_DEEP : (gelu & 16) ? GV_PHASED : (gelu & 32) ? GV_PC : (gelu & 64) ? GV_PHASED192 : GV_AUTO;
    gelu &= 1;
    return gemm(gelu ? EP_GELU : (R ? EP_RESID : 0), (const h16*)A, K, (const h16*)Wt, K, bias, (const h16*)R, N, (h16*)C, N, M, N, K, scale,
                (hipStream_t)stream, nullptr, 0, nullptr, variant);
}

extern "C" int foho_geo_attention(const void* Q, const void* KV, void* Vt_scratch, void* O, int32_t M, int32_t n_latents, int32_t heads,
                                  void* stream) {
    if (!Q || !KV || !Vt_scratch || !O || heads <= 0 || n_latents <= 0 || n_latents % 64 || (size_t)n_latents * heads * 256 >= ((size_t)1 << 31))
        return fail(FOHO_ERR_BAD_ARG, "foho_geo_attention: bad argument");
    const int W = heads * 64;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_geo_pack_vt, dim3((W + 255) / 256, n_latents / 16), dim3(256), 0, s, (const h16*)KV, 2 * W, W, n_latents, (h16*)Vt_scratch);
    if (!launch_ok("k_geo_pack_vt")) return FOHO_ERR_LAUNCH;
    if (M <= 0) return FOHO_OK;
    launch_attn(dim3(((M + AQ - 1) / AQ) * heads), s, (const h16*)Q, W, (const h16*)KV, 2 * W, (const h16*)Vt_scratch, n_latents, (h16*)O, W, M, heads, (float*)nullptr,
                (const int*)nullptr);
    return launch_ok("k_geo_attn") ? FOHO_OK : FOHO_ERR_LAUNCH;
}

#include "foho_vae.inc"
