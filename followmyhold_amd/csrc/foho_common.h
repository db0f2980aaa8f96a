// foho_common.h -- device helpers shared by the gfx950 kernels of libfoho_hip.so.
//
// Everything here is written for CDNA4: 64-lane wavefronts (hard-coded), LDS atomics, wave
// ballots.  The translation units are compiled with -ffp-contract=off so that the geometric
// predicates are plain IEEE-754 binary32 operations in a fixed association order: face indices
// then agree bit-for-bit with the naive per-pixel algorithm of the reference's rasteriser
// (pytorch3d rasterize_meshes naive path; reference src/foho/guidance/run.py:95-105).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FOHO_WAVE 64
#define K_EPS 1e-8f

namespace foho {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---------------------------------------------------------------- wave / block reductions
// Cross-lane reductions use DPP moves (VALU speed) instead of ds_bpermute shuffles, whose ~100-cycle LDS-crossbar
// round trip per step made a 12-value block reduction cost microseconds: butterfly inside each row of 16 lanes
// (quad_perm, row_ror), then row_bcast:15 / row_bcast:31 fold the four rows into lane 63, which is broadcast with
// v_readlane.  Lanes disabled by EXEC or by the row mask contribute the identity.  Results are wave-uniform.
template <int CTRL, int RMASK>
__device__ __forceinline__ int dpp_mov(int ident, int x) {
    return __builtin_amdgcn_update_dpp(ident, x, CTRL, RMASK, 0xf, false);
}
#define FOHO_DPP_REDUCE(T, x, ident, OP, MOV)                     \
    do {                                                          \
        T t_;                                                     \
        t_ = MOV(0xB1, 0xf, ident, x); x = OP(x, t_); /* quad_perm [1,0,3,2] */ \
        t_ = MOV(0x4E, 0xf, ident, x); x = OP(x, t_); /* quad_perm [2,3,0,1] */ \
        t_ = MOV(0x124, 0xf, ident, x); x = OP(x, t_); /* row_ror:4 */          \
        t_ = MOV(0x128, 0xf, ident, x); x = OP(x, t_); /* row_ror:8 */          \
        t_ = MOV(0x142, 0xa, ident, x); x = OP(x, t_); /* row_bcast:15 -> rows 1, 3 */ \
        t_ = MOV(0x143, 0xc, ident, x); x = OP(x, t_); /* row_bcast:31 -> rows 2, 3 */ \
    } while (0)

#define FOHO_MOV_F(C, R, id, x) __int_as_float(dpp_mov<C, R>(__float_as_int(id), __float_as_int(x)))
#define FOHO_MOV_U(C, R, id, x) (unsigned)dpp_mov<C, R>((int)(id), (int)(x))
#define FOHO_OP_ADD(a, b) ((a) + (b))
#define FOHO_OP_MINF(a, b) fminf(a, b)
#define FOHO_OP_MAXF(a, b) fmaxf(a, b)
#define FOHO_OP_MAXU(a, b) (((a) > (b)) ? (a) : (b))

__device__ __forceinline__ float lane63(float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63)); }

__device__ __forceinline__ float wave_sum(float v) {
    FOHO_DPP_REDUCE(float, v, 0.0f, FOHO_OP_ADD, FOHO_MOV_F);
    return lane63(v);
}
__device__ __forceinline__ float wave_min(float v) {
    FOHO_DPP_REDUCE(float, v, INFINITY, FOHO_OP_MINF, FOHO_MOV_F);
    return lane63(v);
}
__device__ __forceinline__ float wave_max(float v) {
    FOHO_DPP_REDUCE(float, v, -INFINITY, FOHO_OP_MAXF, FOHO_MOV_F);
    return lane63(v);
}
__device__ __forceinline__ float wave_sum_all(float v) { return wave_sum(v); }
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
    FOHO_DPP_REDUCE(unsigned, v, 0u, FOHO_OP_ADD, FOHO_MOV_U);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_max(unsigned v) {
    FOHO_DPP_REDUCE(unsigned, v, 0u, FOHO_OP_MAXU, FOHO_MOV_U);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// Block j of a role's n blocks -> the part of the role it takes, such that the blocks an XCD receives (those with equal
// j mod 8: workgroups go to the XCDs round robin) take a contiguous range: c = j mod 8 gets [c*q + min(c, r), ...) with
// q = n / 8, r = n mod 8 -- q + 1 parts for c < r, q otherwise.  A bijection of [0, n).
__device__ __forceinline__ int xcd_order(int j, int n) {
    const int c = j & 7, q = n >> 3, r = n & 7;
    return c * q + min(c, r) + (j >> 3);
}
// inclusive prefix sum over the 64 lanes of a wave (DPP: shifts inside each row of 16, then row_bcast:15 adds row r's
// total to row r+1 for rows 1 and 3, row_bcast:31 adds the total of the first 32 lanes to the last 32).  Whole wave.
__device__ __forceinline__ unsigned wave_incl_scan(unsigned x) {
    x += (unsigned)dpp_mov<0x111, 0xf>(0, (int)x);  // row_shr:1
    x += (unsigned)dpp_mov<0x112, 0xf>(0, (int)x);  // row_shr:2
    x += (unsigned)dpp_mov<0x114, 0xf>(0, (int)x);  // row_shr:4
    x += (unsigned)dpp_mov<0x118, 0xf>(0, (int)x);  // row_shr:8
    x += (unsigned)dpp_mov<0x142, 0xa>(0, (int)x);  // row_bcast:15 -> rows 1, 3
    x += (unsigned)dpp_mov<0x143, 0xc>(0, (int)x);  // row_bcast:31 -> rows 2, 3
    return x;
}

// doubles move as two 32-bit halves
template <int CTRL, int RMASK>
__device__ __forceinline__ double dpp_mov_d(double ident, double x) {
    const long long xi = __double_as_longlong(x), ii = __double_as_longlong(ident);
    const unsigned lo = (unsigned)dpp_mov<CTRL, RMASK>((int)(unsigned)ii, (int)(unsigned)xi);
    const unsigned hi = (unsigned)dpp_mov<CTRL, RMASK>((int)(unsigned)(ii >> 32), (int)(unsigned)(xi >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#define FOHO_MOV_D(C, R, id, x) dpp_mov_d<C, R>(id, x)
__device__ __forceinline__ double wave_sum(double v) {
    FOHO_DPP_REDUCE(double, v, 0.0, FOHO_OP_ADD, FOHO_MOV_D);
    const long long vi = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)vi, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(vi >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Inclusive segmented sum inside each row of 16 lanes (DPP row_shr); `head` = 1 on the first lane of a segment.
// Afterwards the LAST lane of every segment holds the segment total.  Whole wave must call it.
template <int N>
__device__ __forceinline__ void row_segscan(float (&v)[N], int head) {
    int flag = head;
#define FOHO_SEG_STEP(CTRL)                                                                             \
    {                                                                                                   \
        const int fl_d = dpp_mov<CTRL, 0xf>(1, flag); /* lanes without a source see a segment boundary */ \
        _Pragma("unroll") for (int k = 0; k < N; k++) {                                                 \
            const float t = __int_as_float(dpp_mov<CTRL, 0xf>(0, __float_as_int(v[k])));                \
            if (!flag) v[k] += t;                                                                       \
        }                                                                                               \
        flag |= fl_d;                                                                                   \
    }
    FOHO_SEG_STEP(0x111)  // row_shr:1
    FOHO_SEG_STEP(0x112)  // row_shr:2
    FOHO_SEG_STEP(0x114)  // row_shr:4
    FOHO_SEG_STEP(0x118)  // row_shr:8
#undef FOHO_SEG_STEP
}

// block-wide sum of NV floats per thread for 256-thread workgroups; result valid in thread 0.  `red` = NV * 4
// floats of LDS.  Everything is unrolled over compile-time bounds so that thread 0's LDS reads are issued
// back-to-back (a loop over a run-time wave count serialised NV * 4 dependent ds_read round trips).
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
    constexpr int NW = 4;
    const int wv = wave_id();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const float s = wave_sum(v[k]);
        if (lane_id() == 0) red[k * NW + wv] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[NV][NW];
#pragma unroll
        for (int k = 0; k < NV; k++)
#pragma unroll
            for (int w = 0; w < NW; w++) t[k][w] = red[k * NW + w];
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = ((t[k][0] + t[k][1]) + t[k][2]) + t[k][3];
    }
    __syncthreads();
}

// sum over the 16 lanes of a DPP row (butterfly: every lane of the row ends up with the row total)
__device__ __forceinline__ float row16_sum(float x) {
    x += __int_as_float(dpp_mov<0xB1, 0xf>(0, __float_as_int(x)));   // quad_perm [1,0,3,2]
    x += __int_as_float(dpp_mov<0x4E, 0xf>(0, __float_as_int(x)));   // quad_perm [2,3,0,1]
    x += __int_as_float(dpp_mov<0x124, 0xf>(0, __float_as_int(x)));  // row_ror:4
    x += __int_as_float(dpp_mov<0x128, 0xf>(0, __float_as_int(x)));  // row_ror:8
    return x;
}

// Block-wide sums of NV values for 256-thread workgroups, result in LDS: out[k] (k < NV) after the call.  Four DPP
// steps reduce inside each row of 16 lanes, the 16 row totals go through LDS and lane k adds them up -- the
// row_bcast / readlane tail of a full wave reduction costs more than the LDS round trip when NV is large.
// `red` = NV * 16 floats.
template <int NV>
__device__ __forceinline__ void block_sum_lds(const float (&v)[NV], float* red, float* out) {
    const int row = threadIdx.x >> 4;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        const float s = row16_sum(v[k]);
        if ((threadIdx.x & 15) == 0) red[k * 16 + row] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const float* r = red + threadIdx.x * 16;
        float t[16];
#pragma unroll
        for (int j = 0; j < 16; j++) t[j] = r[j];
        float s = t[0];
#pragma unroll
        for (int j = 1; j < 16; j++) s += t[j];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// (value, index) arg-min / arg-max; ties keep the lower index (torch.min(dim) on CPU returns the first)
struct ValIdx {
    float v;
    int i;
};
__device__ __forceinline__ ValIdx vi_min(ValIdx a, ValIdx b) { return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ ValIdx vi_max(ValIdx a, ValIdx b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
template <int CTRL, int RMASK>
__device__ __forceinline__ ValIdx dpp_mov_vi(ValIdx ident, ValIdx x) {
    return ValIdx{__int_as_float(dpp_mov<CTRL, RMASK>(__float_as_int(ident.v), __float_as_int(x.v))), dpp_mov<CTRL, RMASK>(ident.i, x.i)};
}
#define FOHO_MOV_VI(C, R, id, x) dpp_mov_vi<C, R>(id, x)
__device__ __forceinline__ ValIdx wave_vi_min(ValIdx a) {
    const ValIdx ident{INFINITY, 0x7fffffff};
    FOHO_DPP_REDUCE(ValIdx, a, ident, vi_min, FOHO_MOV_VI);
    return ValIdx{lane63(a.v), __builtin_amdgcn_readlane(a.i, 63)};
}
__device__ __forceinline__ ValIdx wave_vi_max(ValIdx a) {
    const ValIdx ident{-INFINITY, 0x7fffffff};
    FOHO_DPP_REDUCE(ValIdx, a, ident, vi_max, FOHO_MOV_VI);
    return ValIdx{lane63(a.v), __builtin_amdgcn_readlane(a.i, 63)};
}

// order-preserving float <-> uint32 (for atomicMin/atomicMax on signed floats)
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}

// ---------------------------------------------------------------- rasteriser primitives
// pixel index -> NDC of the pixel centre (pytorch3d PixToNonSquareNdc).  The per-axis constants are the
// same for every pixel of a launch: they are worked out once on the host (same IEEE binary32 operations) and travel in
// the kernel arguments, so a pixel costs a convert, two multiplies and two adds -- inside per-lane loops the compiler
// would otherwise redo the two uniform divisions on the vector unit every time (float division has no scalar form).
struct PixAxis {
    float range, offset, s1, inv_s1, inv_range;
    int S1;
    int pow2, range_pow2;  // S1 / range are powers of two: dividing by them = multiplying by the exact reciprocal
};
__host__ __device__ __forceinline__ PixAxis pix_axis(int S1, int S2) {
    PixAxis a;
    float range = 2.0f;
    if (S1 > S2) range = ((float)S1 * range) / (float)S2;
    a.range = range;
    a.offset = range / 2.0f;
    a.s1 = (float)S1;
    a.inv_s1 = 1.0f / (float)S1;
    a.inv_range = 1.0f / range;
    a.S1 = S1;
    a.pow2 = (S1 & (S1 - 1)) == 0;
    int e = 0;
    a.range_pow2 = frexpf(range, &e) == 0.5f;
    return a;
}
__device__ __forceinline__ float pix_to_ndc(int i, const PixAxis& a) {
    const float num = a.range * (float)i + a.offset;
    const float q = a.pow2 ? num * a.inv_s1 : num / a.s1;
    return -a.offset + q;
}
__device__ __forceinline__ float pix_to_ndc(int i, int S1, int S2) { return pix_to_ndc(i, pix_axis(S1, S2)); }

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

__device__ __forceinline__ float seg_d2(float px, float py, float ax, float ay, float bx, float by) {
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    if (l2 <= K_EPS) {
        const float dx = px - bx, dy = py - by;
        return dx * dx + dy * dy;
    }
    float t = (bax * (px - ax) + bay * (py - ay)) / l2;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float qx = ax + t * bax, qy = ay + t * bay;
    const float dx = qx - px, dy = qy - py;
    return dx * dx + dy * dy;
}

struct Frag {
    float z, sdist, c0, c1, c2;
};

// One (pixel centre, face) pair of the naive rasteriser with perspective-correct, clipped
// barycentrics (FoV camera + blur_radius > 0 defaults).  fv = 9 floats (x,y,z)x3.
// PRETESTED = true: the caller has established the three face-level tests -- depth not all negative, pixel centre inside
// the blur-inflated box, area beyond K_EPS -- itself (the scatter rasteriser's set-up wave does, per face, and its pixel
// boxes are exactly the centres that pass the box test): same result, ~25 instructions per fragment less.
// full_t > 0 (the scatter rasteriser only): a fragment INSIDE the face whose squared distance to each of the three edge
// LINES exceeds full_t -- e_i^2 > full_t |edge_i|^2, e_i the edge function = |edge_i| x distance to the line -- is farther
// than that from every edge SEGMENT too, so with full_t = 20 sigma its 1 - sigmoid(dist / sigma) is exactly 0 (the float
// sigmoid is 1 from 16.7 on): the pixel is fully covered, the caller needs neither the distance nor its three correctly
// rounded divisions, and out.sdist = -INFINITY says so.  The classification is the one the full evaluation makes: the
// rounding of e_i (<= 2.4e-7 |p - a| |edge_i|, i.e. <= 3e-4 relative at the threshold for edges up to 1 NDC -- longer
// ones never take the shortcut) and of seg_d2 (1e-3) are far inside the margin between 16.7 and 20.
template <bool PRETESTED = false>
__device__ __forceinline__ bool eval_frag(const float* __restrict__ fv, float xf, float yf, float blur_radius,
                                          float sqrt_blur, Frag& out, float full_t = 0.0f) {
    const float x0 = fv[0], y0 = fv[1], z0 = fv[2];
    const float x1 = fv[3], y1 = fv[4], z1 = fv[5];
    const float x2 = fv[6], y2 = fv[7], z2 = fv[8];
    if (!PRETESTED) {
        const float zmax = fmaxf(fmaxf(z0, z1), z2);
        if (zmax < 0.0f) return false;
        const float xmin = fminf(fminf(x0, x1), x2) - sqrt_blur;
        const float xmax = fmaxf(fmaxf(x0, x1), x2) + sqrt_blur;
        const float ymin = fminf(fminf(y0, y1), y2) - sqrt_blur;
        const float ymax = fmaxf(fmaxf(y0, y1), y2) + sqrt_blur;
        if (!(xmin <= xf && xf <= xmax && ymin <= yf && yf <= ymax)) return false;
        const float face_area = edge_fn(x0, y0, x1, y1, x2, y2);
        if (face_area <= K_EPS && face_area >= -K_EPS) return false;
    }

    const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
    const float e0 = edge_fn(xf, yf, x1, y1, x2, y2), e1 = edge_fn(xf, yf, x2, y2, x0, y0), e2 = edge_fn(xf, yf, x0, y0, x1, y1);
    const float a0 = e0 / area;
    const float a1 = e1 / area;
    const float a2 = e2 / area;
    const float t0 = a0 * z1 * z2;
    const float t1 = z0 * a1 * z2;
    const float t2 = z0 * z1 * a2;
    const float den = fmaxf(t0 + t1 + t2, K_EPS);
    const float w0 = t0 / den, w1 = t1 / den, w2 = t2 / den;
    float c0 = fmaxf(w0, 0.0f), c1 = fmaxf(w1, 0.0f), c2 = fmaxf(w2, 0.0f);
    const float s = fmaxf(c0 + c1 + c2, 1e-5f);
    c0 = c0 / s;
    c1 = c1 / s;
    c2 = c2 / s;
    const float pz = c0 * z0 + c1 * z1 + c2 * z2;
    if (pz < 0.0f) return false;
    if (full_t > 0.0f) {
        // e0 belongs to edge (v1, v2), e1 to (v2, v0), e2 to (v0, v1)
        const float l0 = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1), l1 = (x0 - x2) * (x0 - x2) + (y0 - y2) * (y0 - y2),
                    l2 = (x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0);
        if ((w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f) && fmaxf(fmaxf(l0, l1), l2) <= 1.0f && e0 * e0 > full_t * l0 && e1 * e1 > full_t * l1 &&
            e2 * e2 > full_t * l2) {
            out.z = pz + 0.0f;
            out.sdist = -INFINITY;
            out.c0 = c0;
            out.c1 = c1;
            out.c2 = c2;
            return true;
        }
    }
    const float d01 = seg_d2(xf, yf, x0, y0, x1, y1);
    const float d02 = seg_d2(xf, yf, x0, y0, x2, y2);
    const float d12 = seg_d2(xf, yf, x1, y1, x2, y2);
    const float dist = fminf(fminf(d01, d02), d12);
    const bool inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
    if (!inside && dist >= blur_radius) return false;
    out.z = pz + 0.0f;  // canonicalise -0 -> +0 so the uint ordering of the z key holds
    out.sdist = inside ? -dist : dist;
    out.c0 = c0;
    out.c1 = c1;
    out.c2 = c2;
    return true;
}

// Signed squared edge distance of a fragment that eval_frag ACCEPTED (the winner k_resolve decodes from the z key): the
// same three seg_d2 values and minimum, bit for bit, and the inside test without eval_frag's nine other divisions.
// inside <=> w_i = t_i / max(t0 + t1 + t2, eps) > 0 for all i with t_0 = a_0 z_1 z_2, a_i = e_i / area: the divisor is
// positive and neither quotient can underflow to zero (|e_i| >= ~1e-22 where it is not exactly zero, |area| <= 4, depths
// <= zfar), so sign(w_i) = sign(e_i) sign(area) sign(z_j z_k) -- compared as products, never divided.
__device__ __forceinline__ float winner_sdist(const float* __restrict__ fv, float xf, float yf) {
    const float x0 = fv[0], y0 = fv[1], z0 = fv[2];
    const float x1 = fv[3], y1 = fv[4], z1 = fv[5];
    const float x2 = fv[6], y2 = fv[7], z2 = fv[8];
    const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
    const float sa = (area > 0.0f) ? 1.0f : ((area < 0.0f) ? -1.0f : 0.0f);
    const float s0 = (edge_fn(xf, yf, x1, y1, x2, y2) * sa) * (z1 * z2);
    const float s1 = (edge_fn(xf, yf, x2, y2, x0, y0) * sa) * (z0 * z2);
    const float s2 = (edge_fn(xf, yf, x0, y0, x1, y1) * sa) * (z0 * z1);
    const float d01 = seg_d2(xf, yf, x0, y0, x1, y1);
    const float d02 = seg_d2(xf, yf, x0, y0, x2, y2);
    const float d12 = seg_d2(xf, yf, x1, y1, x2, y2);
    const float dist = fminf(fminf(d01, d02), d12);
    const bool inside = (s0 > 0.0f) && (s1 > 0.0f) && (s2 > 0.0f);
    return inside ? -dist : dist;
}

// ------------------------------------------------------------------------------------------------
// Near plane.  MeshRasterizer hands z_clip_value = znear / 2 to rasterize_meshes for perspective cameras (reference
// src/foho/guidance/run.py:84-105, z_clip_value=None), which runs pytorch3d's clip_faces on the (x_ndc, y_ndc, z_view)
// face vertices first: a face with all three vertices nearer than the plane is dropped, a face that straddles it is
// replaced by ONE triangle (two vertices behind: (p4, p5, p1), p1 the vertex in front) or by TWO (one vertex behind:
// (p4, p2, p5) and (p5, p2, p3), p1 the vertex behind; p2, p3 follow p1 in the face's cyclic order; p4 / p5 = where the
// edges p1p2 / p1p3 cross the plane, w = (z1 - c) / (z1 - z_o), z = z1 (1 - w) + z_o w, xy = ((xy1 z1)(1 - w) +
// (xy_o z_o) w) / c for a perspective camera).  The two halves of a split face never both leave a fragment on a pixel: the
// second replaces the first when its unsigned edge distance is smaller (rasterize_meshes, CheckPixelInsideFace).
// Everything below is selects and straight-line arithmetic on scalars (run-time indexed private arrays would live in
// scratch memory); it only runs for faces that straddle the plane -- none in the path's working range, the hand and the
// object sit decimetres in front of the camera.
// ------------------------------------------------------------------------------------------------
struct ClipGeom {
    int n;        // sub-triangles: 0 = culled, 1, 2
    int i1;       // position of p1 in the face
    float w2, w3; // crossing weights on p1p2 / p1p3
};
__device__ __forceinline__ void sel_rot(const float* fv, int i1, float* p1, float* p2, float* p3) {
    // values first, selects second: selecting between the LOADS lets the optimiser turn them into one load with a run-time
    // index, and a run-time indexed private array is moved to LDS / scratch memory
    const bool r1 = i1 == 1, r2 = i1 == 2;
    _Pragma("unroll") for (int q = 0; q < 3; q++) {
        const float a = fv[q], b = fv[3 + q], c = fv[6 + q];
        p1[q] = r1 ? b : (r2 ? c : a);
        p2[q] = r1 ? c : (r2 ? a : b);
        p3[q] = r1 ? a : (r2 ? b : c);
    }
}
__device__ __forceinline__ void plane_crossing(const float* p1, const float* po, float c, float& w, float* out) {
    w = (p1[2] - c) / (p1[2] - po[2]);
    const float u = 1.0f - w;
    out[0] = ((p1[0] * p1[2]) * u + (po[0] * po[2]) * w) / c;
    out[1] = ((p1[1] * p1[2]) * u + (po[1] * po[2]) * w) / c;
    out[2] = p1[2] * u + po[2] * w;
}
// sub-triangles of a face with at least one vertex nearer than c (the caller checks that): t0, t1 (t1 only when n == 2)
__device__ __forceinline__ ClipGeom clip_subtris(const float* fv, float c, float* t0, float* t1) {
    const bool b0 = fv[2] < c, b1 = fv[5] < c, b2 = fv[8] < c;
    const int nb = (int)b0 + (int)b1 + (int)b2;
    ClipGeom g;
    g.n = (nb == 3 || nb == 0) ? 0 : ((nb == 2) ? 1 : 2);
    g.i1 = (nb == 2) ? (!b0 ? 0 : (!b1 ? 1 : 2)) : (b0 ? 0 : (b1 ? 1 : 2));
    g.w2 = g.w3 = 0.0f;
    if (g.n == 0) return g;
    float p1[3], p2[3], p3[3], p4[3], p5[3];
    sel_rot(fv, g.i1, p1, p2, p3);
    plane_crossing(p1, p2, c, g.w2, p4);
    plane_crossing(p1, p3, c, g.w3, p5);
    _Pragma("unroll") for (int q = 0; q < 3; q++) {
        t0[q] = p4[q];
        t0[3 + q] = (g.n == 1) ? p5[q] : p2[q];
        t0[6 + q] = (g.n == 1) ? p1[q] : p5[q];
        t1[q] = p5[q];
        t1[3 + q] = p2[q];
        t1[6 + q] = p3[q];
    }
    return g;
}
// eval_frag for a face that may straddle the near plane; sub = -1 (not clipped), 0 or 1 (the sub-triangle the surviving
// fragment belongs to).  Barycentrics in `out` are the sub-triangle's.
__device__ __forceinline__ bool eval_frag_near(const float* __restrict__ fv, float zc, float xf, float yf, float blur_radius,
                                               float sqrt_blur, Frag& out, int& sub) {
    sub = -1;
    if (!(fminf(fminf(fv[2], fv[5]), fv[8]) < zc)) return eval_frag(fv, xf, yf, blur_radius, sqrt_blur, out);
    float t0[9], t1[9];
    const ClipGeom g = clip_subtris(fv, zc, t0, t1);
    if (g.n == 0) return false;
    Frag f0, f1;
    const bool r0 = eval_frag(t0, xf, yf, blur_radius, sqrt_blur, f0);
    bool r1 = false;
    if (g.n == 2) r1 = eval_frag(t1, xf, yf, blur_radius, sqrt_blur, f1);
    if (!r0 && !r1) return false;
    const bool second = r1 && (!r0 || fabsf(f1.sdist) < fabsf(f0.sdist));  // the neighbour replaces only when strictly closer
    sub = second ? 1 : 0;
    out = second ? f1 : f0;
    return true;
}
// barycentrics of a sub-triangle fragment -> barycentrics w.r.t. the unclipped face (pytorch3d
// convert_clipped_rasterization_to_original_faces): rows of the conversion = barycentrics of the sub-triangle's vertices
__device__ __forceinline__ void subtri_bary_to_face(const float* fv, float zc, int sub, float* b) {
    float t0[9], t1[9];
    const ClipGeom g = clip_subtris(fv, zc, t0, t1);
    // barycentrics of p4, p5 in (p1, p2, p3) order: p4 = (1 - w2, w2, 0), p5 = (1 - w3, 0, w3)
    float r[3];  // weights of p1, p2, p3
    if (g.n == 1) {  // (p4, p5, p1)
        r[0] = b[0] * (1.0f - g.w2) + b[1] * (1.0f - g.w3) + b[2];
        r[1] = b[0] * g.w2;
        r[2] = b[1] * g.w3;
    } else if (sub == 0) {  // (p4, p2, p5)
        r[0] = b[0] * (1.0f - g.w2) + b[2] * (1.0f - g.w3);
        r[1] = b[0] * g.w2 + b[1];
        r[2] = b[2] * g.w3;
    } else {  // (p5, p2, p3)
        r[0] = b[0] * (1.0f - g.w3);
        r[1] = b[1];
        r[2] = b[0] * g.w3 + b[2];
    }
    // p1, p2, p3 sit at positions i1, i1 + 1, i1 + 2 (mod 3) of the face
    b[0] = (g.i1 == 0) ? r[0] : ((g.i1 == 1) ? r[2] : r[1]);
    b[1] = (g.i1 == 0) ? r[1] : ((g.i1 == 1) ? r[0] : r[2]);
    b[2] = (g.i1 == 0) ? r[2] : ((g.i1 == 1) ? r[1] : r[0]);
}

// d(seg_d2)/d(a,b) with the projection parameter held constant (envelope; pytorch3d
// PointLineDistanceBackward).  Accumulates g * d(dist)/d(.) into ga[2], gb[2].
// a / b through v_rcp_f32 (1 ulp) for GRADIENT arithmetic only: parity there is 1e-4 relative, and the correctly
// rounded division costs ~10 instructions.  Everything that decides a face index, a depth or a tie stays exact.
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

__device__ __forceinline__ void seg_d2_bwd(float px, float py, float ax, float ay, float bx, float by, float g,
                                           float* ga, float* gb) {
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    if (l2 <= K_EPS) {
        gb[0] += g * 2.0f * (bx - px);
        gb[1] += g * 2.0f * (by - py);
        return;
    }
    float t = fdiv(bax * (px - ax) + bay * (py - ay), l2);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float dx = (ax + t * bax) - px, dy = (ay + t * bay) - py;
    ga[0] += g * 2.0f * dx * (1.0f - t);
    ga[1] += g * 2.0f * dy * (1.0f - t);
    gb[0] += g * 2.0f * dx * t;
    gb[1] += g * 2.0f * dy * t;
}

// Backward of eval_frag for one fragment: inputs dL/dz (g_z), dL/dbary_clip (g_c[3]), dL/dsdist (g_sd);
// accumulates into gv[9] = dL/d(x0,y0,z0,x1,y1,z1,x2,y2,z2).  
__device__ __forceinline__ void eval_frag_bwd(const float* __restrict__ fv, float xf, float yf, float g_z,
                                              const float* g_cin, float g_sd, float* gv) {
    const float x0 = fv[0], y0 = fv[1], z0 = fv[2];
    const float x1 = fv[3], y1 = fv[4], z1 = fv[5];
    const float x2 = fv[6], y2 = fv[7], z2 = fv[8];
    const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
    const float e0 = edge_fn(xf, yf, x1, y1, x2, y2), e1 = edge_fn(xf, yf, x2, y2, x0, y0),
                e2 = edge_fn(xf, yf, x0, y0, x1, y1);
    const float a0 = fdiv(e0, area), a1 = fdiv(e1, area), a2 = fdiv(e2, area);
    const float t0 = a0 * z1 * z2, t1 = z0 * a1 * z2, t2 = z0 * z1 * a2;
    const float tsum = t0 + t1 + t2;
    const float den = fmaxf(tsum, K_EPS);
    const float w0 = fdiv(t0, den), w1 = fdiv(t1, den), w2 = fdiv(t2, den);
    const float cp0 = fmaxf(w0, 0.0f), cp1 = fmaxf(w1, 0.0f), cp2 = fmaxf(w2, 0.0f);
    const float csum = cp0 + cp1 + cp2;
    const float s = fmaxf(csum, 1e-5f);
    const float c0 = fdiv(cp0, s), c1 = fdiv(cp1, s), c2 = fdiv(cp2, s);
    const bool inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);

    if (g_z != 0.0f || g_cin[0] != 0.0f || g_cin[1] != 0.0f || g_cin[2] != 0.0f) {
        // pz = c . z
        const float gc0 = g_cin[0] + g_z * z0, gc1 = g_cin[1] + g_z * z1, gc2 = g_cin[2] + g_z * z2;
        float gz0 = g_z * c0, gz1 = g_z * c1, gz2 = g_z * c2;
        // c = cp / s
        float gcp0, gcp1, gcp2;
        if (csum >= 1e-5f) {
            const float dot = gc0 * c0 + gc1 * c1 + gc2 * c2;
            gcp0 = fdiv(gc0 - dot, s);
            gcp1 = fdiv(gc1 - dot, s);
            gcp2 = fdiv(gc2 - dot, s);
        } else {
            gcp0 = fdiv(gc0, s);
            gcp1 = fdiv(gc1, s);
            gcp2 = fdiv(gc2, s);
        }
        const float gw0 = (w0 >= 0.0f) ? gcp0 : 0.0f, gw1 = (w1 >= 0.0f) ? gcp1 : 0.0f, gw2 = (w2 >= 0.0f) ? gcp2 : 0.0f;
        // w = t / den
        float gt0, gt1, gt2;
        if (tsum >= K_EPS) {
            const float dot = gw0 * w0 + gw1 * w1 + gw2 * w2;
            gt0 = fdiv(gw0 - dot, den);
            gt1 = fdiv(gw1 - dot, den);
            gt2 = fdiv(gw2 - dot, den);
        } else {
            gt0 = fdiv(gw0, den);
            gt1 = fdiv(gw1, den);
            gt2 = fdiv(gw2, den);
        }
        const float ga0 = gt0 * z1 * z2, ga1 = gt1 * z0 * z2, ga2 = gt2 * z0 * z1;
        gz1 += gt0 * a0 * z2;
        gz2 += gt0 * a0 * z1;
        gz0 += gt1 * a1 * z2;
        gz2 += gt1 * z0 * a1;
        gz0 += gt2 * z1 * a2;
        gz1 += gt2 * z0 * a2;
        const float ge0 = fdiv(ga0, area), ge1 = fdiv(ga1, area), ge2 = fdiv(ga2, area);
        const float garea = -fdiv(ga0 * a0 + ga1 * a1 + ga2 * a2, area);
        // e0 = E(p; v1, v2)
        gv[3] += ge0 * (yf - y2);
        gv[4] += ge0 * (x2 - xf);
        gv[6] += -ge0 * (yf - y1);
        gv[7] += ge0 * (xf - x1);
        // e1 = E(p; v2, v0)
        gv[6] += ge1 * (yf - y0);
        gv[7] += ge1 * (x0 - xf);
        gv[0] += -ge1 * (yf - y2);
        gv[1] += ge1 * (xf - x2);
        // e2 = E(p; v0, v1)
        gv[0] += ge2 * (yf - y1);
        gv[1] += ge2 * (x1 - xf);
        gv[3] += -ge2 * (yf - y0);
        gv[4] += ge2 * (xf - x0);
        // area = E(v2; v0, v1) + eps
        gv[6] += garea * (y1 - y0);
        gv[7] += -garea * (x1 - x0);
        gv[0] += garea * (y2 - y1);
        gv[1] += garea * (x1 - x2);
        gv[3] += -garea * (y2 - y0);
        gv[4] += garea * (x2 - x0);
        gv[2] += gz0;
        gv[5] += gz1;
        gv[8] += gz2;
    }
    if (g_sd != 0.0f) {
        const float gd = inside ? -g_sd : g_sd;
        const float d01 = seg_d2(xf, yf, x0, y0, x1, y1);
        const float d02 = seg_d2(xf, yf, x0, y0, x2, y2);
        const float d12 = seg_d2(xf, yf, x1, y1, x2, y2);
        // nearest edge: 0 = (v0,v1), 1 = (v0,v2), 2 = (v1,v2).  End points and destinations are chosen with selects and
        // constant indices -- handing seg_d2_bwd a run-time pointer into gv[] would move the array to scratch memory.
        const int sel = (d01 <= d02 && d01 <= d12) ? 0 : ((d02 <= d01 && d02 <= d12) ? 1 : 2);
        const float ax = (sel == 2) ? x1 : x0, ay = (sel == 2) ? y1 : y0;
        const float bx = (sel == 0) ? x1 : x2, by = (sel == 0) ? y1 : y2;
        float ga[2] = {0.f, 0.f}, gb[2] = {0.f, 0.f};
        seg_d2_bwd(xf, yf, ax, ay, bx, by, gd, ga, gb);
        gv[0] += (sel != 2) ? ga[0] : 0.0f;
        gv[1] += (sel != 2) ? ga[1] : 0.0f;
        gv[3] += (sel == 0) ? gb[0] : ((sel == 2) ? ga[0] : 0.0f);
        gv[4] += (sel == 0) ? gb[1] : ((sel == 2) ? ga[1] : 0.0f);
        gv[6] += (sel != 0) ? gb[0] : 0.0f;
        gv[7] += (sel != 0) ? gb[1] : 0.0f;
    }
}

// gradient of a crossing point p4 = f(p1, po) (plane_crossing) pushed back onto p1 and po
__device__ __forceinline__ void plane_crossing_bwd(const float* p1, const float* po, float c, float w, const float* g4, float* g1,
                                                   float* go) {
    const float u = 1.0f - w, ic = 1.0f / c;
    g1[0] += g4[0] * p1[2] * u * ic;
    g1[1] += g4[1] * p1[2] * u * ic;
    go[0] += g4[0] * po[2] * w * ic;
    go[1] += g4[1] * po[2] * w * ic;
    const float gw = (g4[0] * (po[0] * po[2] - p1[0] * p1[2]) + g4[1] * (po[1] * po[2] - p1[1] * p1[2])) * ic + g4[2] * (po[2] - p1[2]);
    const float dz = p1[2] - po[2], idz2 = 1.0f / (dz * dz);
    g1[2] += (g4[0] * p1[0] + g4[1] * p1[1]) * u * ic + g4[2] * u + gw * (c - po[2]) * idz2;
    go[2] += (g4[0] * po[0] + g4[1] * po[1]) * w * ic + g4[2] * w + gw * (p1[2] - c) * idz2;
}
// eval_frag_bwd for a face that may straddle the near plane: the fragment's sub-triangle is found again with the forward
// rule, differentiated, and its vertex gradients are pushed through the cut onto the face's own vertices (the cut moves
// with them).  g_cin refers to the SUB-TRIANGLE's barycentrics (the fused step never sends gradient into barycentrics).
__device__ __forceinline__ void eval_frag_near_bwd(const float* __restrict__ fv, float zc, float blur_radius, float sqrt_blur,
                                                   float xf, float yf, float g_z, const float* g_cin, float g_sd, float* gv) {
    // One straight path, no early returns: stores into gv[] from several exits get merged by the optimiser into ONE store
    // through a selected pointer, i.e. a run-time index, which moves the caller's accumulator array to scratch memory.
    const bool near = fminf(fminf(fv[2], fv[5]), fv[8]) < zc;
    float t0[9], t1[9];
    _Pragma("unroll") for (int q = 0; q < 9; q++) t0[q] = t1[q] = fv[q];
    ClipGeom g;
    g.n = 1;
    g.i1 = 0;
    g.w2 = g.w3 = 0.5f;
    bool second = false, live = true;
    if (near) {
        g = clip_subtris(fv, zc, t0, t1);
        Frag f0, f1;
        const bool r0 = g.n > 0 && eval_frag(t0, xf, yf, blur_radius, sqrt_blur, f0);
        const bool r1 = g.n == 2 && eval_frag(t1, xf, yf, blur_radius, sqrt_blur, f1);
        second = r1 && (!r0 || fabsf(f1.sdist) < fabsf(f0.sdist));
        live = r0 || r1;
    }
    float ts[9], gt[9];
    _Pragma("unroll") for (int q = 0; q < 9; q++) {
        ts[q] = second ? t1[q] : t0[q];
        gt[q] = 0.0f;
    }
    eval_frag_bwd(ts, xf, yf, live ? g_z : 0.0f, g_cin, live ? g_sd : 0.0f, gt);
    float d[9];
    _Pragma("unroll") for (int q = 0; q < 9; q++) d[q] = gt[q];
    if (near) {
        float p1[3], p2[3], p3[3], g1[3] = {0.f, 0.f, 0.f}, g2[3] = {0.f, 0.f, 0.f}, g3[3] = {0.f, 0.f, 0.f};
        float g4[3], g5[3];
        sel_rot(fv, g.i1, p1, p2, p3);
        const bool one = g.n == 1;
        _Pragma("unroll") for (int q = 0; q < 3; q++) {
            // (p4, p5, p1) | (p4, p2, p5) | (p5, p2, p3)
            g4[q] = (one || !second) ? gt[q] : 0.0f;
            g5[q] = one ? gt[3 + q] : (second ? gt[q] : gt[6 + q]);
            g1[q] = one ? gt[6 + q] : 0.0f;
            g2[q] = one ? 0.0f : gt[3 + q];
            g3[q] = (!one && second) ? gt[6 + q] : 0.0f;
        }
        plane_crossing_bwd(p1, p2, zc, g.w2, g4, g1, g2);
        plane_crossing_bwd(p1, p3, zc, g.w3, g5, g1, g3);
        const bool r1 = g.i1 == 1, r2 = g.i1 == 2;
        _Pragma("unroll") for (int q = 0; q < 3; q++) {  // p1, p2, p3 back to positions i1, i1 + 1, i1 + 2 (mod 3)
            const float a = g1[q], b = g2[q], c = g3[q];
            d[q] = r1 ? c : (r2 ? b : a);
            d[3 + q] = r1 ? a : (r2 ? c : b);
            d[6 + q] = r1 ? b : (r2 ? a : c);
        }
    }
    _Pragma("unroll") for (int q = 0; q < 9; q++) gv[q] += d[q];
}

// sigmoid in fp32 (torch.sigmoid: 1 / (1 + exp(-x)))
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// quaternion wxyz -> R (row-major 9), pytorch3d quaternion_to_matrix (valid for non-unit q)
__device__ __forceinline__ void quat_to_mat(const float* q, float* R) {
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float two_s = 2.0f / (((r * r + i * i) + j * j) + k * k);
    R[0] = 1.0f - two_s * (j * j + k * k);
    R[1] = two_s * (i * j - k * r);
    R[2] = two_s * (i * k + j * r);
    R[3] = two_s * (i * j + k * r);
    R[4] = 1.0f - two_s * (i * i + k * k);
    R[5] = two_s * (j * k - i * r);
    R[6] = two_s * (i * k - j * r);
    R[7] = two_s * (j * k + i * r);
    R[8] = 1.0f - two_s * (i * i + j * j);
}

}  // namespace foho
