// foho_step.hip -- the batched guidance step on MI355X (gfx950).
//
// One call of foho_step_run() = one optimisation iteration of the reference's phase A / B / C inner
// loops (third_party_patches/hy3dgen/shapegen/pipelines.py:1320-1358, 1386-1453, 1480-1601) for a
// batch of B independent images, as a short chain of kernels on one HIP stream with no host
// synchronisation:
//
//   k_xform        similarity transforms about the AABB centre + FoV projection       (PL:108-118, 242-250)
//   k_stage2       vertex normals | face setup + 16x16-px binning | K=1 NN | keypoints | edge/verts^2
//   k_raster       per-tile z-buffer in LDS (ds_min_u64 keys), silhouette product      (RUN:95-116)
//   k_loss         normal / disparity / BCE partial sums, min-max path sums, tie counts (PL:272-289, 178-186)
//   k_stats        reduce the partials of one render
//   k_face_bwd     face-centric backward of shading + rasteriser (no atomics)
//   k_frac_bwd     silhouette backward over the fractional-coverage fragment list
//   k_vert_gather  faces -> vertices (CSR), vertex-normal backward part 1
//   k_vert_bwd     vertex-normal backward part 2, projection backward, similarity partial sums
//   k_inside_*     +z ray parity of the joint (res+1)^3 grid via atomicXor on column bit masks (SDF:131-160)
//   k_final        loss assembly, parameter gradients, Adam/AdamW update                (PL:1578-1601)
//
// Data layout in HBM: vertices AoS (V,3) f32, faces (F,3) i32 global ids, per-face NDC copy (F,9) f32 and
// pixel box (F,4) i16 written once per step by the face-setup role; G-buffer per render = 4 planes
// (face id i32, z f32, signed dist f32, silhouette product f32) = 16 B/px.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/foho_hip.h"
#include "foho_common.h"

using namespace foho;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
void foho_set_error(const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* foho_last_error(void) { return g_err; }
extern "C" int foho_version(void) { return 100; }

#define CHECK_LAUNCH(name)                                   \
    do {                                                     \
        hipError_t e_ = hipGetLastError();                   \
        if (e_ != hipSuccess) {                              \
            char b_[200];                                    \
            snprintf(b_, sizeof(b_), "%s: %s", name, hipGetErrorString(e_)); \
            foho_set_error(b_);                              \
            return FOHO_ERR_LAUNCH;                          \
        }                                                    \
    } while (0)

// ------------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------------
constexpr int TILE = 16;              // raster tile edge (px); one 256-thread workgroup per tile
constexpr int BIN_CAP = 2048;         // face ids per tile bin; the rest spills to the per-render global list
constexpr int BIN_MAX_TILES = 64;     // faces touching more tiles than this go to the global list
constexpr int SMALL_AREA = 48;        // bbox-in-tile area (px) up to which a face is rasterised by one lane
constexpr int LARGE_Q = 512;          // LDS queue of large faces per tile
constexpr int K_SIL = 100;            // faces_per_pixel of the silhouette rasteriser (RUN:109)
constexpr int LOSS_BLOCKS = 64;       // blocks of the per-pixel loss pass per (render, image)
constexpr int NPART = 12;             // partial sums per loss block
constexpr int VERT_BLOCKS_MAX = 1024;  // blocks of vertex-role partials per image
constexpr int SIM_NP = 20;            // similarity-backward partial sums per block
constexpr int NSTAT = 32;             // finalised per-render stats (floats)

struct MeshInfo {  // per (image, mesh)
    float center[3];
    int argmin[3];
    int argmax[3];
    float tmin[3];  // AABB of the transformed vertices (SDF grid)
    float tmax[3];
    int pad;
};

struct FracEntry {
    int pix;
    int face;
    float sdist;
};

// raw stats accumulated by hit tiles of the rasteriser
struct RStats {
    unsigned hit_count;
    unsigned rgb_min_inv, rgb_max;    // ordered-uint encoded over hit pixels' 3 channels; minima are stored
    unsigned disp_min_inv, disp_max;  // bit-inverted and accumulated with atomicMax so that 0 = "empty"
    unsigned flags;               // bit1 frac overflow, bit2 >K fragments on a pixel
    unsigned glob_count;          // entries of the global (spill) face list
    unsigned pad;
};

// ------------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------------
struct WS {
    size_t total;
    size_t world, ndc, vn_raw, vn, mesh_info, face_ndc, face_box;
    size_t p2f, zbuf, sdist, prod;
    size_t bin_count, bin_list, glob_list;
    size_t frac, frac_count, rstats, loss_part, stats2;
    size_t face_gcol, face_gndc, g_n, g_ndc, g_raw, g_world, g_direct;
    size_t knn_idx, knn_d2, kp3d, g_kp3d, vert_part, sim_part, parity, int_count;
    int tiles_x, tiles_y, ntiles;
    size_t zero_begin, zero_end;  // region cleared by one memset per step
};

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

static WS make_ws(const foho_dims& d) {
    WS w;
    size_t o = 0;
    const size_t P = (size_t)d.H * d.W, R = d.n_renders, B = d.B;
    const size_t V3 = (size_t)d.Vtot * 3 * 4;
    w.tiles_x = (d.W + TILE - 1) / TILE;
    w.tiles_y = (d.H + TILE - 1) / TILE;
    w.ntiles = w.tiles_x * w.tiles_y;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = al(o + bytes);
        return r;
    };
    // --- zeroed every step (atomic accumulators) ---
    w.zero_begin = o;
    w.bin_count = take(R * B * w.ntiles * 4);
    w.frac_count = take(R * B * 4);
    w.rstats = take(R * B * sizeof(RStats));
    w.g_world = take(V3);
    const int G1 = d.grid_res + 1;
    w.parity = take(B * 2 * (size_t)G1 * G1 * 16);
    w.int_count = take(B * 4);
    w.zero_end = o;
    // --- plain scratch ---
    w.world = take(V3);
    w.ndc = take(V3);
    w.vn_raw = take(V3);
    w.vn = take(V3);
    w.mesh_info = take(B * 2 * sizeof(MeshInfo));
    w.face_ndc = take((size_t)d.Ftot * 9 * 4);
    w.face_box = take((size_t)d.Ftot * 8);
    w.p2f = take(R * B * P * 4);
    w.zbuf = take(R * B * P * 4);
    w.sdist = take(R * B * P * 4);
    w.prod = take(R * B * P * 4);
    w.bin_list = take(R * B * (size_t)w.ntiles * BIN_CAP * 4);
    w.glob_list = take(R * B * (size_t)d.Fmax * 4);
    w.frac = take(R * B * (size_t)d.frac_cap * sizeof(FracEntry));
    w.loss_part = take(R * B * LOSS_BLOCKS * NPART * 4);
    w.stats2 = take(R * B * NSTAT * 4);
    w.face_gcol = take(R * (size_t)d.Ftot * 3 * 4);
    w.face_gndc = take(R * (size_t)d.Ftot * 9 * 4);
    w.g_n = take(V3);
    w.g_ndc = take(V3);
    w.g_raw = take(V3);
    w.g_direct = take(V3);
    w.knn_idx = take((size_t)d.Vtot * 4);
    w.knn_d2 = take((size_t)d.Vtot * 4);
    w.kp3d = take(B * 21 * 3 * 4);
    w.g_kp3d = take(B * 21 * 3 * 4);
    w.vert_part = take(B * VERT_BLOCKS_MAX * 8 * 4);
    w.sim_part = take(B * 2 * VERT_BLOCKS_MAX * SIM_NP * 4);
    w.total = o;
    return w;
}

extern "C" size_t foho_step_workspace_bytes(const foho_dims* dims) {
    if (!dims) return 0;
    return make_ws(*dims).total;
}

extern "C" int64_t foho_step_workspace_region(const foho_dims* dims, int region, int64_t* nbytes) {
    if (!dims) return -1;
    const WS w = make_ws(*dims);
    const foho_dims& d = *dims;
    const size_t P = (size_t)d.H * d.W, R = d.n_renders, B = d.B, V3 = (size_t)d.Vtot * 12;
    const int G1 = d.grid_res + 1;
    size_t off, n;
    switch (region) {
        case FOHO_WS_WORLD: off = w.world; n = V3; break;
        case FOHO_WS_NDC: off = w.ndc; n = V3; break;
        case FOHO_WS_VN: off = w.vn; n = V3; break;
        case FOHO_WS_P2F: off = w.p2f; n = R * B * P * 4; break;
        case FOHO_WS_ZBUF: off = w.zbuf; n = R * B * P * 4; break;
        case FOHO_WS_SDIST: off = w.sdist; n = R * B * P * 4; break;
        case FOHO_WS_PROD: off = w.prod; n = R * B * P * 4; break;
        case FOHO_WS_KNN_IDX: off = w.knn_idx; n = (size_t)d.Vtot * 4; break;
        case FOHO_WS_KNN_D2: off = w.knn_d2; n = (size_t)d.Vtot * 4; break;
        case FOHO_WS_GWORLD: off = w.g_world; n = V3; break;
        case FOHO_WS_FRAC_COUNT: off = w.frac_count; n = R * B * 4; break;
        case FOHO_WS_STATS: off = w.stats2; n = R * B * NSTAT * 4; break;
        case FOHO_WS_PARITY: off = w.parity; n = B * 2 * (size_t)G1 * G1 * 16; break;
        default: return -1;
    }
    if (nbytes) *nbytes = (int64_t)n;
    return (int64_t)off;
}

// kernel-side view of the step (device pointers, by value)
struct Ctx {
    foho_dims d;
    const foho_image* img;
    const float* verts_in;
    const int32_t* faces;
    const int32_t *inc_off, *inc_fc, *nbr_off, *nbr_idx;
    const float* J;
    const float *tgt_normal, *tgt_disp;
    const uint8_t* mask;
    const float* kps_2d;
    float *params, *adam_m, *adam_v;
    int32_t* adam_t;
    float *losses, *grad_params, *grad_verts_in;
    int32_t* flags;
    // workspace
    float *world, *ndc, *vn_raw, *vn;
    MeshInfo* mesh_info;
    float* face_ndc;
    short4* face_box;
    int32_t* p2f;
    float *zbuf, *sdist, *prod;
    unsigned* bin_count;
    int32_t *bin_list, *glob_list;
    FracEntry* frac;
    unsigned* frac_count;
    RStats* rstats;
    float *loss_part, *stats2;
    float *face_gcol, *face_gndc, *g_n, *g_ndc, *g_raw, *g_world, *g_direct;
    int32_t* knn_idx;
    float *knn_d2, *kp3d, *g_kp3d, *vert_part, *sim_part;
    unsigned long long* parity;
    int32_t* int_count;
    int tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ void face_range(const foho_image& im, int face_set, int& f0, int& f1) {
    if (face_set == FOHO_FACES_HAND) {
        f0 = im.f_off;
        f1 = im.f_off + im.Fh;
    } else if (face_set == FOHO_FACES_OBJ) {
        f0 = im.f_off + im.Fh;
        f1 = im.f_off + im.Fh + im.Fo;
    } else {
        f0 = im.f_off;
        f1 = im.f_off + im.Fh + im.Fo;
    }
}

// ------------------------------------------------------------------------------------------------
// k_xform: grid (2 meshes, B), 1024 threads.  One workgroup walks one mesh three times: AABB of the
// input (centre; differentiable through the arg-min/max vertices, PL:111), transform + project,
// AABB of the output (SDF grid, SDF:138-144).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_moge(const Ctx& c, const foho_image& im, int mesh, int gi, float* v) {
    const float x = c.verts_in[3 * gi], y = c.verts_in[3 * gi + 1], z = c.verts_in[3 * gi + 2];
    if (mesh == 0) {
        v[0] = x;
        v[1] = y;
        v[2] = z;
    } else {  // transform_hunyuan2moge (PL:242-250)
        const float* M = im.T_h2m;
        v[0] = ((x * M[0] + y * M[1]) + z * M[2]) + M[3];
        v[1] = ((x * M[4] + y * M[5]) + z * M[6]) + M[7];
        v[2] = ((x * M[8] + y * M[9]) + z * M[10]) + M[11];
    }
}

__global__ __launch_bounds__(1024) void k_xform(Ctx c) {
    const int mesh = blockIdx.x, b = blockIdx.y;
    const foho_image im = c.img[b];
    const int base = im.v_off + (mesh ? im.Vh : 0);
    const int n = mesh ? im.Vo : im.Vh;
    MeshInfo* mi = &c.mesh_info[b * 2 + mesh];
    __shared__ ValIdx s_mn[3][16], s_mx[3][16];
    __shared__ float s_center[3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    ValIdx mn[3], mx[3];
    for (int k = 0; k < 3; k++) {
        mn[k] = {INFINITY, 0x7fffffff};
        mx[k] = {-INFINITY, 0x7fffffff};
    }
    for (int i = tid; i < n; i += blockDim.x) {
        float v[3];
        load_moge(c, im, mesh, base + i, v);
        for (int k = 0; k < 3; k++) {
            mn[k] = vi_min(mn[k], ValIdx{v[k], i});
            mx[k] = vi_max(mx[k], ValIdx{v[k], i});
        }
    }
    for (int k = 0; k < 3; k++) {
        ValIdx a = wave_vi_min(mn[k]), m2 = wave_vi_max(mx[k]);
        if (lane == 0) {
            s_mn[k][wv] = a;
            s_mx[k][wv] = m2;
        }
    }
    __syncthreads();
    if (tid < 3) {
        ValIdx a = s_mn[tid][0], m2 = s_mx[tid][0];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
            a = vi_min(a, s_mn[tid][w]);
            m2 = vi_max(m2, s_mx[tid][w]);
        }
        const float cen = (a.v + m2.v) / 2.0f;
        s_center[tid] = cen;
        mi->center[tid] = cen;
        mi->argmin[tid] = a.i;
        mi->argmax[tid] = m2.i;
    }
    __syncthreads();
    const float cx = s_center[0], cy = s_center[1], cz = s_center[2];
    const float* p = c.params + b * 16 + mesh * 8;  // [s, t(3), q(4)]
    const float s = p[0], tx = p[1], ty = p[2], tz = p[3];
    float Rm[9];
    quat_to_mat(p + 4, Rm);
    float tmn[3] = {INFINITY, INFINITY, INFINITY}, tmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += blockDim.x) {
        float v[3];
        load_moge(c, im, mesh, base + i, v);
        const float u0 = s * (v[0] - cx), u1 = s * (v[1] - cy), u2 = s * (v[2] - cz);
        const float wx = ((u0 * Rm[0] + u1 * Rm[1]) + u2 * Rm[2] + cx) + tx;
        const float wy = ((u0 * Rm[3] + u1 * Rm[4]) + u2 * Rm[5] + cy) + ty;
        const float wz = ((u0 * Rm[6] + u1 * Rm[7]) + u2 * Rm[8] + cz) + tz;
        const int gi = base + i;
        c.world[3 * gi] = wx;
        c.world[3 * gi + 1] = wy;
        c.world[3 * gi + 2] = wz;
        const float* Rc = im.cam_R;
        const float vx = ((wx * Rc[0] + wy * Rc[3]) + wz * Rc[6]) + im.cam_T[0];
        const float vy = ((wx * Rc[1] + wy * Rc[4]) + wz * Rc[7]) + im.cam_T[1];
        const float vz = ((wx * Rc[2] + wy * Rc[5]) + wz * Rc[8]) + im.cam_T[2];
        c.ndc[3 * gi] = (im.k00 * vx) / vz;
        c.ndc[3 * gi + 1] = (im.k11 * vy) / vz;
        c.ndc[3 * gi + 2] = vz;
        tmn[0] = fminf(tmn[0], wx);
        tmn[1] = fminf(tmn[1], wy);
        tmn[2] = fminf(tmn[2], wz);
        tmx[0] = fmaxf(tmx[0], wx);
        tmx[1] = fmaxf(tmx[1], wy);
        tmx[2] = fmaxf(tmx[2], wz);
    }
    __syncthreads();
    float* sf = reinterpret_cast<float*>(&s_mn[0][0]);  // reuse LDS: 6 x 16 floats
    for (int k = 0; k < 3; k++) {
        const float a = wave_min(tmn[k]), m2 = wave_max(tmx[k]);
        if (lane == 0) {
            sf[k * 16 + wv] = a;
            sf[(3 + k) * 16 + wv] = m2;
        }
    }
    __syncthreads();
    if (tid < 3) {
        float a = sf[tid * 16], m2 = sf[(3 + tid) * 16];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
            a = fminf(a, sf[tid * 16 + w]);
            m2 = fmaxf(m2, sf[(3 + tid) * 16 + w]);
        }
        mi->tmin[tid] = a;
        mi->tmax[tid] = m2;
    }
}

// ------------------------------------------------------------------------------------------------
// k_stage2: multi-role launch over everything that only depends on k_xform's output.
// grid.x = nA + nB + nC + nD + nE blocks of 256 threads, grid.y = B.
//   A vertex normals   B face setup + tile binning   C K=1 NN hand->object (+contact)
//   D 21 keypoints     E object edge loss + verts^2 (loss partials and direct gradients)
// ------------------------------------------------------------------------------------------------
struct Stage2Cfg {
    int nA, nB, nC, nD, nE;
    foho_render_cfg render[2];
    int n_renders;
    float sqrt_blur;
    float w_edge, w_verts, w_contact, contact_margin;
};

__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ void role_normals(const Ctx& c, const foho_image& im, int blk) {
    const int i = blk * 256 + threadIdx.x;
    if (i >= im.Vh + im.Vo) return;
    const int gv = im.v_off + i;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int e = c.inc_off[gv]; e < c.inc_off[gv + 1]; e++) {
        const int f = c.inc_fc[e] >> 2;
        const int i0 = c.faces[3 * f], i1 = c.faces[3 * f + 1], i2 = c.faces[3 * f + 2];
        float A[3], Bv[3], n[3];
        for (int k = 0; k < 3; k++) {
            A[k] = c.world[3 * i2 + k] - c.world[3 * i1 + k];
            Bv[k] = c.world[3 * i0 + k] - c.world[3 * i1 + k];
        }
        cross3(A, Bv, n);
        acc[0] += n[0];
        acc[1] += n[1];
        acc[2] += n[2];
    }
    const float nrm = sqrtf((acc[0] * acc[0] + acc[1] * acc[1]) + acc[2] * acc[2]);
    const float den = fmaxf(nrm, 1e-6f);
    for (int k = 0; k < 3; k++) {
        c.vn_raw[3 * gv + k] = acc[k];
        c.vn[3 * gv + k] = acc[k] / den;
    }
}

// conservative pixel-index range of pixel centres inside [lo, hi] (NDC), for an axis of S1 pixels
__device__ __forceinline__ void ndc_to_pix_range(float lo, float hi, int S1, int S2, int& p0, int& p1) {
    float range = 2.0f;
    if (S1 > S2) range = ((float)S1 * range) / (float)S2;
    const float off = range / 2.0f;
    // pixel centre of flipped index i: -off + (range*i + off)/S1  ->  i = ((x + off)*S1 - off)/range
    float ilo = ((lo + off) * (float)S1 - off) / range;
    float ihi = ((hi + off) * (float)S1 - off) / range;
    ilo = fminf(fmaxf(ilo, -2.0f), (float)S1 + 1.0f);
    ihi = fminf(fmaxf(ihi, -2.0f), (float)S1 + 1.0f);
    const int i0 = (int)floorf(ilo) - 1, i1 = (int)ceilf(ihi) + 1;
    p0 = max(0, S1 - 1 - i1);  // unflipped pixel index
    p1 = min(S1 - 1, S1 - 1 - i0);
}

__device__ void role_face_setup(const Ctx& c, const foho_image& im, const Stage2Cfg& cfg, int b, int blk) {
    const int lf = blk * 256 + threadIdx.x;
    if (lf >= im.Fh + im.Fo) return;
    const int f = im.f_off + lf;
    float fv[9];
    for (int k = 0; k < 3; k++) {
        const int vi = c.faces[3 * f + k];
        fv[3 * k] = c.ndc[3 * vi];
        fv[3 * k + 1] = c.ndc[3 * vi + 1];
        fv[3 * k + 2] = c.ndc[3 * vi + 2];
    }
    for (int k = 0; k < 9; k++) c.face_ndc[9 * (size_t)f + k] = fv[k];
    const int H = c.d.H, W = c.d.W;
    const float zmax = fmaxf(fmaxf(fv[2], fv[5]), fv[8]);
    const float farea = edge_fn(fv[0], fv[1], fv[3], fv[4], fv[6], fv[7]);
    bool valid = !(zmax < 0.0f) && !(farea <= K_EPS && farea >= -K_EPS);
    int x0 = 1, x1 = 0, y0 = 1, y1 = 0;
    if (valid) {
        const float xlo = fminf(fminf(fv[0], fv[3]), fv[6]) - cfg.sqrt_blur;
        const float xhi = fmaxf(fmaxf(fv[0], fv[3]), fv[6]) + cfg.sqrt_blur;
        const float ylo = fminf(fminf(fv[1], fv[4]), fv[7]) - cfg.sqrt_blur;
        const float yhi = fmaxf(fmaxf(fv[1], fv[4]), fv[7]) + cfg.sqrt_blur;
        valid = (xlo == xlo) && (xhi == xhi) && (ylo == ylo) && (yhi == yhi);
        if (valid) {
            ndc_to_pix_range(xlo, xhi, W, H, x0, x1);
            ndc_to_pix_range(ylo, yhi, H, W, y0, y1);
            valid = (x0 <= x1) && (y0 <= y1);
        }
    }
    short4 box;
    box.x = (short)(valid ? x0 : 1);
    box.y = (short)(valid ? x1 : 0);
    box.z = (short)(valid ? y0 : 1);
    box.w = (short)(valid ? y1 : 0);
    c.face_box[f] = box;
    if (!valid) return;
    const int tx0 = x0 / TILE, tx1 = x1 / TILE, ty0 = y0 / TILE, ty1 = y1 / TILE;
    const int nt = (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
    const bool is_hand = lf < im.Fh;
    for (int r = 0; r < cfg.n_renders; r++) {
        const int fs = cfg.render[r].face_set;
        if ((fs == FOHO_FACES_HAND && !is_hand) || (fs == FOHO_FACES_OBJ && is_hand)) continue;
        const size_t rb = (size_t)r * c.d.B + b;
        unsigned* bc = c.bin_count + rb * c.ntiles;
        int32_t* bl = c.bin_list + rb * (size_t)c.ntiles * BIN_CAP;
        if (nt <= BIN_MAX_TILES) {
            // binned into every tile it touches; a full bin drops the id but keeps counting, which tells
            // k_raster to fall back to scanning the whole face range for that tile.
            for (int ty = ty0; ty <= ty1; ty++)
                for (int tx = tx0; tx <= tx1; tx++) {
                    const int t = ty * c.tiles_x + tx;
                    const unsigned slot = atomicAdd(&bc[t], 1u);
                    if (slot < BIN_CAP) bl[(size_t)t * BIN_CAP + slot] = f;
                }
        } else {  // screen-filling face: one entry in the per-render global list, visited by every tile
            const unsigned slot = atomicAdd(&c.rstats[rb].glob_count, 1u);
            c.glob_list[rb * (size_t)c.d.Fmax + slot] = f;
        }
    }
}

__device__ void role_knn(const Ctx& c, const foho_image& im, const Stage2Cfg& cfg, int b, int blk, float* red) {
    // one wave per hand vertex; lanes stride over the object vertices
    const int wv = wave_id(), lane = lane_id();
    const int i = blk * 4 + wv;
    float contact = 0.f, d2v = 0.f;
    if (i < im.Vh && im.Vo > 0) {
        const int gi = im.v_off + i;
        const float px = c.world[3 * gi], py = c.world[3 * gi + 1], pz = c.world[3 * gi + 2];
        ValIdx best{INFINITY, 0x7fffffff};
        const int ob = im.v_off + im.Vh;
        for (int j = lane; j < im.Vo; j += 64) {
            const float dx = px - c.world[3 * (ob + j)], dy = py - c.world[3 * (ob + j) + 1],
                        dz = pz - c.world[3 * (ob + j) + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            best = vi_min(best, ValIdx{d, j});
        }
        best = wave_vi_min(best);
        if (lane == 0) {
            c.knn_idx[gi] = best.i;
            c.knn_d2[gi] = best.v;
            d2v = best.v;
            const float a = best.v - cfg.contact_margin;
            contact = fmaxf(a, 0.0f);
            if (a >= 0.0f && cfg.w_contact != 0.0f) {  // clamp(min=0) passes the gradient at equality
                const int gj = ob + best.i;
                const float g = cfg.w_contact / (float)im.Vh;
                const float dx = px - c.world[3 * gj], dy = py - c.world[3 * gj + 1], dz = pz - c.world[3 * gj + 2];
                atomicAdd(&c.g_world[3 * gj], -2.0f * dx * g);
                atomicAdd(&c.g_world[3 * gj + 1], -2.0f * dy * g);
                atomicAdd(&c.g_world[3 * gj + 2], -2.0f * dz * g);
            }
        }
    }
    // per-block partials: slot = blk within role C
    if (lane == 0) {
        red[wv] = contact;
        red[4 + wv] = d2v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* vp = c.vert_part + ((size_t)b * VERT_BLOCKS_MAX + blk) * 8;
        vp[0] = (red[0] + red[1]) + (red[2] + red[3]);
        vp[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

__constant__ int c_tips[5] = {744, 320, 443, 554, 671};  // PL:127
__constant__ int c_kp_order[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};  // PL:128

__device__ void role_kps(const Ctx& c, const foho_image& im, int b, int blk) {
    // one wave per output keypoint slot (21 slots -> 6 blocks of 4 waves)
    const int slot = blk * 4 + wave_id(), lane = lane_id();
    if (slot >= 21) return;
    const int src = c_kp_order[slot];
    float acc[3] = {0.f, 0.f, 0.f};
    if (src < 16) {
        const float* Jr = c.J + (size_t)src * im.jcols;
        for (int v = lane; v < im.jcols; v += 64) {
            const float w = Jr[v];
            const int gv = im.v_off + v;
            acc[0] += w * c.world[3 * gv];
            acc[1] += w * c.world[3 * gv + 1];
            acc[2] += w * c.world[3 * gv + 2];
        }
        for (int k = 0; k < 3; k++) acc[k] = wave_sum(acc[k]);
    } else if (lane == 0) {
        const int gv = im.v_off + c_tips[src - 16];
        for (int k = 0; k < 3; k++) acc[k] = c.world[3 * gv + k];
    }
    if (lane == 0)
        for (int k = 0; k < 3; k++) c.kp3d[((size_t)b * 21 + slot) * 3 + k] = acc[k];
}

__device__ void role_obj_local(const Ctx& c, const foho_image& im, const Stage2Cfg& cfg, int b, int blk, float* red) {
    const int i = blk * 256 + threadIdx.x;
    float v[2] = {0.f, 0.f};  // edge sum (unique edges), verts^2 sum
    if (i < im.Vo) {
        const int gv = im.v_off + im.Vh + i;
        const float x = c.world[3 * gv], y = c.world[3 * gv + 1], z = c.world[3 * gv + 2];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        for (int e = c.nbr_off[gv]; e < c.nbr_off[gv + 1]; e++) {
            const int j = c.nbr_idx[e];
            const float dx = x - c.world[3 * j], dy = y - c.world[3 * j + 1], dz = z - c.world[3 * j + 2];
            if (j > gv) v[0] += dx * dx + dy * dy + dz * dz;
            gx += dx;
            gy += dy;
            gz += dz;
        }
        v[1] = x * x + y * y + z * z;
        const float ke = (im.n_edges > 0) ? cfg.w_edge * 2.0f / (float)im.n_edges : 0.0f;
        const float kv = cfg.w_verts * 2.0f / (3.0f * (float)im.Vo);
        c.g_direct[3 * gv] = ke * gx + kv * x;
        c.g_direct[3 * gv + 1] = ke * gy + kv * y;
        c.g_direct[3 * gv + 2] = ke * gz + kv * z;
    }
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        float* vp = c.vert_part + ((size_t)b * VERT_BLOCKS_MAX + blk) * 8;
        vp[2] = v[0];
        vp[3] = v[1];
    }
}

__global__ __launch_bounds__(256) void k_stage2(Ctx c, Stage2Cfg cfg) {
    __shared__ float red[32];
    const int b = blockIdx.y;
    const foho_image im = c.img[b];
    int blk = blockIdx.x;
    if (blk < cfg.nA) return role_normals(c, im, blk);
    blk -= cfg.nA;
    if (blk < cfg.nB) return role_face_setup(c, im, cfg, b, blk);
    blk -= cfg.nB;
    if (blk < cfg.nC) return role_knn(c, im, cfg, b, blk, red);
    blk -= cfg.nC;
    if (blk < cfg.nD) return role_kps(c, im, b, blk);
    blk -= cfg.nD;
    if (blk < cfg.nE) return role_obj_local(c, im, cfg, b, blk, red);
}

// softmax_rgb_blend with K=1 on the colour n_a + n_b + n_c (PL:82-92).  Shared by the
// rasteriser epilogue, the loss pass and the backward pass so that `rgb == max` tests see identical bits.
struct Shade {
    float rgb[3], col[3], p, wnum, delta, den;
};

__device__ __forceinline__ void shade_pixel(const Ctx& c, const foho_image& im, size_t gf, float z, float sd, float sigma,
                                            float gamma, Shade& s) {
    const int i0 = c.faces[3 * gf], i1 = c.faces[3 * gf + 1], i2 = c.faces[3 * gf + 2];
    s.p = sigmoidf(-sd / sigma);
    const float z_inv = (im.zfar - z) / (im.zfar - im.znear);
    const float z_inv_max = fmaxf(z_inv, 1e-10f);
    s.wnum = s.p * expf((z_inv - z_inv_max) / gamma);
    s.delta = fmaxf(expf((1e-10f - z_inv_max) / gamma), 1e-10f);
    s.den = s.wnum + s.delta;
    for (int k = 0; k < 3; k++) {
        s.col[k] = (c.vn[3 * i0 + k] + c.vn[3 * i1 + k]) + c.vn[3 * i2 + k];
        s.rgb[k] = (s.wnum * s.col[k] + s.delta * 1.0f) / s.den;
    }
}

// ------------------------------------------------------------------------------------------------
// k_raster: one 256-thread workgroup per 16x16-px tile of one (render, image).
// LDS holds the tile's z-buffer as 64-bit keys (z bits << 32 | face id): ds_min_u64 implements "nearest
// fragment wins, ties keep the lowest face id" independent of visiting order.  Small faces are rasterised
// face-parallel (one lane walks the face's pixels inside the tile); faces whose in-tile box is large are
// queued in LDS and rasterised pixel-parallel afterwards.  The silhouette product prod_k(1 - p_k) of every
// fragment on the pixel (SoftSilhouetteShader, K=100) is kept in LDS with a CAS multiply; fragments with
// fractional coverage (0 < p < 1) are appended to a global list for the backward pass.
// ------------------------------------------------------------------------------------------------
struct RasterCfg {
    foho_render_cfg render[2];
    float blur_radius, sqrt_blur, sigma, gamma;
};

__device__ __forceinline__ void lds_mul(float* addr, float f) {
    unsigned* ua = reinterpret_cast<unsigned*>(addr);
    unsigned old = *ua, assumed;
    do {
        assumed = old;
        old = atomicCAS(ua, assumed, __float_as_uint(__uint_as_float(assumed) * f));
    } while (old != assumed);
}

__device__ __forceinline__ void raster_pixel_face(const Ctx& c, const RasterCfg& cfg, const float* fv, int f, int px,
                                                  int py, int lx, int ly, unsigned long long* s_key, float* s_prod,
                                                  unsigned* s_cnt, size_t rb, int pix_index) {
    const int H = c.d.H, W = c.d.W;
    const float yf = pix_to_ndc(H - 1 - py, H, W);
    const float xf = pix_to_ndc(W - 1 - px, W, H);
    Frag fr;
    if (!eval_frag(fv, xf, yf, cfg.blur_radius, cfg.sqrt_blur, fr)) return;
    const int li = ly * TILE + lx;
    const unsigned long long key = ((unsigned long long)__float_as_uint(fr.z) << 32) | (unsigned)f;
    atomicMin(&s_key[li], key);
    atomicAdd(&s_cnt[li], 1u);
    const float p = sigmoidf(-fr.sdist / cfg.sigma);
    const float om = 1.0f - p;
    if (om == 0.0f) {
        atomicAnd(reinterpret_cast<unsigned*>(&s_prod[li]), 0u);
    } else {
        lds_mul(&s_prod[li], om);
        const unsigned slot = atomicAdd(&c.frac_count[rb], 1u);
        if (slot < (unsigned)c.d.frac_cap) {
            FracEntry e{pix_index, f, fr.sdist};
            c.frac[rb * (size_t)c.d.frac_cap + slot] = e;
        } else {
            atomicOr(&c.rstats[rb].flags, 2u);
        }
    }
}

__global__ __launch_bounds__(256) void k_raster(Ctx c, RasterCfg cfg) {
    __shared__ unsigned long long s_key[TILE * TILE];
    __shared__ float s_prod[TILE * TILE];
    __shared__ unsigned s_cnt[TILE * TILE];
    __shared__ int s_large[LARGE_Q];
    __shared__ unsigned s_nlarge;
    __shared__ float s_red[4 * 4];
    __shared__ unsigned s_hits[4];

    const int tile = blockIdx.x, rbi = blockIdx.y;
    const int r = rbi / c.d.B, b = rbi % c.d.B;
    const size_t rb = rbi;
    const foho_image im = c.img[b];
    const int H = c.d.H, W = c.d.W;
    const int tx = tile % c.tiles_x, ty = tile / c.tiles_x;
    const int X0 = tx * TILE, Y0 = ty * TILE;
    const int X1 = min(X0 + TILE - 1, W - 1), Y1 = min(Y0 + TILE - 1, H - 1);
    const int tid = threadIdx.x;
    int f0r, f1r;
    face_range(im, cfg.render[r].face_set, f0r, f1r);
    const int fbase = f0r;  // face ids reported relative to the render's own mesh

    s_key[tid] = ~0ull;
    s_prod[tid] = 1.0f;
    s_cnt[tid] = 0u;
    if (tid == 0) s_nlarge = 0;
    __syncthreads();

    // Work list of this tile: its bin + the global list, or (bin overflowed) the render's whole face range.
    const unsigned nbin_raw = c.bin_count[rb * c.ntiles + tile];
    const bool brute = nbin_raw > (unsigned)BIN_CAP;
    const int nbin = brute ? (f1r - f0r) : (int)nbin_raw;
    const int nglob = brute ? 0 : (int)c.rstats[rb].glob_count;
    const int32_t* bl = c.bin_list + (rb * (size_t)c.ntiles + tile) * BIN_CAP;
    const int32_t* gl = c.glob_list + rb * (size_t)c.d.Fmax;
    const int ntot = nbin + nglob;
    const size_t pbase = rb * (size_t)H * W;
    auto list_face = [&](int it) -> int { return brute ? (f0r + it) : (it < nbin ? bl[it] : gl[it - nbin]); };

    if (ntot > 0) {
        for (int it = tid; it < ntot; it += 256) {
            const int f = list_face(it);
            const short4 bx = c.face_box[f];
            const int x0 = max((int)bx.x, X0), x1 = min((int)bx.y, X1), y0 = max((int)bx.z, Y0), y1 = min((int)bx.w, Y1);
            if (x0 > x1 || y0 > y1) continue;
            const int area = (x1 - x0 + 1) * (y1 - y0 + 1);
            if (area > SMALL_AREA) {
                const unsigned q = atomicAdd(&s_nlarge, 1u);
                if (q < LARGE_Q) s_large[q] = f;
                continue;
            }
            float fv[9];
            for (int k = 0; k < 9; k++) fv[k] = c.face_ndc[9 * (size_t)f + k];
            for (int py = y0; py <= y1; py++)
                for (int px = x0; px <= x1; px++)
                    raster_pixel_face(c, cfg, fv, f - fbase, px, py, px - X0, py - Y0, s_key, s_prod, s_cnt, rb,
                                      py * W + px);
        }
        __syncthreads();
        // large faces: pixel-parallel
        const unsigned nl = s_nlarge;
        const int lx = tid % TILE, ly = tid / TILE;
        const int px = X0 + lx, py = Y0 + ly;
        if (nl <= LARGE_Q) {
            for (unsigned q = 0; q < nl; q++) {
                const int f = s_large[q];
                const short4 bx = c.face_box[f];
                if (px < W && py < H && px >= bx.x && px <= bx.y && py >= bx.z && py <= bx.w) {
                    float fv[9];
                    for (int k = 0; k < 9; k++) fv[k] = c.face_ndc[9 * (size_t)f + k];
                    raster_pixel_face(c, cfg, fv, f - fbase, px, py, lx, ly, s_key, s_prod, s_cnt, rb, py * W + px);
                }
            }
        } else {  // LDS queue overflowed: rescan the work list for the large faces
            for (int it = 0; it < ntot; it++) {
                const int f = list_face(it);
                const short4 bx = c.face_box[f];
                const int x0 = max((int)bx.x, X0), x1 = min((int)bx.y, X1), y0 = max((int)bx.z, Y0),
                          y1 = min((int)bx.w, Y1);
                if (x0 > x1 || y0 > y1) continue;
                if ((x1 - x0 + 1) * (y1 - y0 + 1) <= SMALL_AREA) continue;
                if (px >= x0 && px <= x1 && py >= y0 && py <= y1) {
                    float fv[9];
                    for (int k = 0; k < 9; k++) fv[k] = c.face_ndc[9 * (size_t)f + k];
                    raster_pixel_face(c, cfg, fv, f - fbase, px, py, lx, ly, s_key, s_prod, s_cnt, rb, py * W + px);
                }
            }
        }
        __syncthreads();
    }

    // resolve: one pixel per thread
    const int lx = tid % TILE, ly = tid / TILE;
    const int px = X0 + lx, py = Y0 + ly;
    float mn = INFINITY, mx = -INFINITY, dmn = INFINITY, dmx = -INFINITY;
    unsigned hit = 0;
    if (px < W && py < H) {
        const size_t pi = pbase + (size_t)py * W + px;
        const unsigned long long key = s_key[tid];
        if (key == ~0ull) {
            c.p2f[pi] = -1;
            c.zbuf[pi] = -1.0f;
            c.sdist[pi] = -1.0f;
            c.prod[pi] = 1.0f;
        } else {
            const int f = (int)(unsigned)(key & 0xffffffffull);  // relative to the render's mesh
            const float z = __uint_as_float((unsigned)(key >> 32));
            float fv[9];
            const size_t gf = (size_t)(f + fbase);
            for (int k = 0; k < 9; k++) fv[k] = c.face_ndc[9 * gf + k];
            const float yf = pix_to_ndc(H - 1 - py, H, W), xf = pix_to_ndc(W - 1 - px, W, H);
            Frag fr;
            eval_frag(fv, xf, yf, cfg.blur_radius, cfg.sqrt_blur, fr);
            c.p2f[pi] = f;
            c.zbuf[pi] = z;
            c.sdist[pi] = fr.sdist;
            c.prod[pi] = s_prod[tid];
            if (s_cnt[tid] > (unsigned)K_SIL) atomicOr(&c.rstats[rb].flags, 4u);
            // colour for the global min/max of render_normal_and_disparity (PL:279, PL:285)
            Shade sh;
            shade_pixel(c, im, gf, z, fr.sdist, cfg.sigma, cfg.gamma, sh);
            for (int k = 0; k < 3; k++) {
                mn = fminf(mn, sh.rgb[k]);
                mx = fmaxf(mx, sh.rgb[k]);
            }
            const float disp = 1.0f / (z + 1e-6f);
            dmn = disp;
            dmx = disp;
            hit = 1;
        }
    }
    const unsigned long long hb = __ballot(hit != 0);
    if (__syncthreads_or(hit != 0)) {
        mn = wave_min(mn);
        mx = wave_max(mx);
        dmn = wave_min(dmn);
        dmx = wave_max(dmx);
        const int wv = wave_id();
        if (lane_id() == 0) {
            s_red[wv] = mn;
            s_red[4 + wv] = mx;
            s_red[8 + wv] = dmn;
            s_red[12 + wv] = dmx;
            s_hits[wv] = (unsigned)__popcll(hb);
        }
        __syncthreads();
        if (tid == 0) {
            RStats* st = &c.rstats[rb];
            atomicAdd(&st->hit_count, s_hits[0] + s_hits[1] + s_hits[2] + s_hits[3]);
            atomicMax(&st->rgb_min_inv, ~f2ord(fminf(fminf(s_red[0], s_red[1]), fminf(s_red[2], s_red[3]))));
            atomicMax(&st->rgb_max, f2ord(fmaxf(fmaxf(s_red[4], s_red[5]), fmaxf(s_red[6], s_red[7]))));
            atomicMax(&st->disp_min_inv, ~f2ord(fminf(fminf(s_red[8], s_red[9]), fminf(s_red[10], s_red[11]))));
            atomicMax(&st->disp_max, f2ord(fmaxf(fmaxf(s_red[12], s_red[13]), fmaxf(s_red[14], s_red[15]))));
        }
    }
}
#include "foho_step_part2.inc"
