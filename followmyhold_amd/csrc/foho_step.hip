// foho_step.hip -- the batched guidance step on MI355X (gfx950).
//
// One call of foho_step_run() = one optimisation iteration of the reference's phase A / B / C inner
// loops (third_party_patches/hy3dgen/shapegen/pipelines.py:1320-1358, 1386-1453, 1480-1601) for a
// batch of B independent images, as a short chain of kernels on one HIP stream with no host
// synchronisation:
//
//   (k_bbox*)      FOHO_STAGE_BBOX, once: AABB of the input meshes = centre of the similarity transform (PL:111)
//   k_xform        similarity transform about the AABB centre + FoV projection (PL:108-118, 242-250); clears
//                  the step's accumulators                                                   [k_vertex.inc]
//   k_stage2       roles: scatter rasteriser (z-key atomics, RUN:95-116) | K=1 NN | inside test (+z ray parity
//                  via atomicXor on column bit masks) | vertex normals | keypoints | edge / verts^2
//                                                                       [k_raster.inc, k_inside.inc, k_vertex.inc]
//   k_resolve      z-keys -> G-buffer, hit-tile flags, colour / disparity min-max              [k_raster.inc]
//   k_loss         normal / disparity / BCE partial sums of all renders, render statistics (last workgroup),
//                  keypoint loss, intersection popcount                                        [k_loss.inc]
//   k_pix_bwd      per-pixel backward of loss heads + shading + rasteriser, reduced per vertex in LDS, and
//                  the silhouette backward over the fractional-coverage fragment list        [k_backward.inc]
//   k_vert_bwd     vertex-normal / projection / contact / keypoint backward, similarity partial sums; the
//                  last workgroup runs the final stage: loss assembly, parameter gradients, Adam/AdamW
//                  (PL:1578-1601)                                              [k_backward.inc, k_final.inc]
//
// Data layout in HBM: vertices AoS (V,3) f32, faces (F,3) i32 global ids, per-face NDC copy (F,9) f32 written by
// the rasteriser's setup; per render scatter planes (z-key u64, full-coverage byte, fractional-fragment counter and
// fixed-point log sum; all-zero between steps) and a G-buffer (face id i32 for every pixel; z, signed dist, silhouette product, colour for
// hit pixels only).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>

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
extern "C" int foho_version(void) { return 105; }
extern "C" int foho_abi_sizes(int64_t out[5]) {
    out[0] = sizeof(foho_image), out[1] = sizeof(foho_dims), out[2] = sizeof(foho_render_cfg), out[3] = sizeof(foho_step_cfg),
    out[4] = sizeof(foho_step_desc);
    return 105;
}

// optional per-kernel timing (foho_step_run_profiled): one hipEvent after every launch
struct ProfState {
    hipEvent_t ev[32];
    const char* name[32];
    int n;
};
static thread_local ProfState* g_prof = nullptr;

#define CHECK_LAUNCH(kname_)                                                                   \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) {                                                              \
            char b_[200];                                                                    \
            snprintf(b_, sizeof(b_), "%s: %s", kname_, hipGetErrorString(e_));                 \
            foho_set_error(b_);                                                              \
            return FOHO_ERR_LAUNCH;                                                          \
        }                                                                                    \
        if (g_prof && g_prof->n < 32) {                                                      \
            g_prof->name[g_prof->n] = kname_;                                                  \
            (void)hipEventRecord(g_prof->ev[g_prof->n++], stream);                           \
        }                                                                                    \
    } while (0)

#include "foho_stamps.h"   // development instrumentation (time stamps, ablation switches): nothing in the product build

// ------------------------------------------------------------------------------------------------
// constants
// ------------------------------------------------------------------------------------------------
constexpr int BTX = 32, BTY = 8;      // pixel tile of k_resolve / k_pix_bwd (one pixel per lane; 32 px = one 128-B line of a 4-B plane)
constexpr int RF = 64;                // faces per workgroup of the scatter rasteriser (= one wave for the setup scan)
constexpr int RQ_CAP = 1024;          // LDS queue of (face slot, pixel) candidates per enumerate round (4 KB: with the staged nearest-neighbour role, k_stage2 needs ~8 KB of LDS per workgroup)
constexpr int FRAC_SEG = 256;          // fractional-coverage fragments a raster workgroup keeps in its own list segment per render
constexpr int K_SIL = 100;            // faces_per_pixel of the silhouette rasteriser (RUN:109)
constexpr int KFIX_MAX = 32;          // pixels per (render, image) whose K-buffer is re-built exactly (k_resolve: kbuffer_fix)
constexpr int KFIX_FRAGS = 1024;      // fragments such a pixel may hold
constexpr int LOSS_BLOCKS = 256;      // blocks of the per-pixel loss pass per (render, image)
constexpr int NPART = 12;             // partial sums per loss block
constexpr int LOSS_SLOTS = 16;        // copies of a render's 12 loss accumulators (same-address f64 atomics serialise: 3 us of tail with one copy)
constexpr int VERT_BLOCKS_MAX = 1024; // blocks of vertex-role partials per image
constexpr int SIM_NP = 20;            // similarity-backward partial sums per workgroup (18 used)
constexpr int VB = 64;                // vertices per workgroup of k_vert_bwd at one image (128 in batches: vert_block)
constexpr int VP_CAP = 1024;          // (vertex, incident face) pairs staged in LDS per round of k_vert_bwd
constexpr int SIM_ROWS_MAX = VERT_BLOCKS_MAX * (256 / VB);  // partial rows per (image, mesh)
constexpr int NSTAT = 32;             // finalised per-render stats (floats)
constexpr int SIM_ACC = 40;            // floats per image and parity in the deferred-update accumulators (36 used)
constexpr int STATE_NEXT = 64;         // floats per image in the deferred-update staging area
constexpr int PIX_BWD_TILE_BLOCKS = 256;
constexpr int BWD_SPLIT = 4;             // a dense 32x8 tile is handed to k_pix_bwd as up to 4 bands of pixel rows  // k_pix_bwd workgroups per (render, image) walking the hit-tile list
constexpr int BWD_SLOTS = 512;        // LDS hash slots: a work-list entry holds at most 160 hit pixels (k_resolve's band split), i.e. <= 480 distinct vertices

struct MeshInfo {  // per (image, mesh): AABB of the INPUT vertices, recomputed by FOHO_STAGE_BBOX only
    unsigned long long kmin_inv[3];  // ~(ordered value << 32 | index), atomicMax  -> min value, lowest index
    unsigned long long kmax[3];      //  (ordered value << 32 | ~index), atomicMax -> max value, lowest index
};

struct FracEntry {
    int pix;
    int face;
    float sdist;
};
struct KFixEntry {  // a pixel with more than K_SIL candidate fragments: only fragments with key <= thr are in its K-buffer
    int pix;
    int pad;
    unsigned long long thr;  // (z bits << 32 | face id) of the farthest fragment kept
};

// raw stats accumulated by the hit workgroups of k_resolve.  Same-address device-scope atomics serialise at a few
// ns each, so the accumulators are spread over NSLOT cache lines (workgroup i -> slot i % NSLOT) and k_loss
// reduces the slots.
constexpr int NSLOT = 64;
struct RSlot {
    unsigned hit_count;
    unsigned rgb_min_inv, rgb_max;    // ordered-uint encoded over hit pixels' 3 channels; minima are stored
    unsigned disp_min_inv, disp_max;  // bit-inverted and accumulated with atomicMax so that 0 = "empty"
    unsigned pad[11];
};
struct RStats {
    unsigned flags;                   // bit1 frac overflow, bit2 >K fragments on a pixel, bit3 face across the near plane culled
    unsigned pad[3];
};

// ------------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------------
struct WS {
    size_t total;
    size_t world, ndc, vn_raw, vn, mesh_info, face_ndc;
    size_t p2f, zbuf, sdist, prod, pcol, hit_list, hit_count, bwd_list, bwd_count, tile_touched, tile_clean, pair_v, pending, state_next, sim_acc;
    size_t zkey, fullb, nfrac, plog;
    size_t kfix_count, kfix;
    size_t clean_begin, clean_end;  // scatter planes: cleared by FOHO_STAGE_BBOX, kept clean by k_resolve
    size_t frac, frac_count, frac_seg, seg_count, rstats, rslot, loss_part, stats2;
    int nseg;
    size_t g_ndc, g_nrm, g_world, g_direct;
    size_t knn_idx, knn_d2, kp3d, g_kp3d, vert_part, sim_part, xf_part, g_special, parity, int_count, final_ticket, knn_inv, loss_acc, vbox, hand_order;
    size_t tile_static, image_static, act_list, act_count;
    int btiles_x, nbtiles;
    size_t zero_begin, zero_end;  // region cleared by k_zero every step
};

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// faces per workgroup of the scatter rasteriser: power of two in [8, RF], about F / 256
static inline int vert_block(const foho_dims& d) { return d.B > 1 ? 2 * VB : VB; }  // vertices per k_vert_bwd workgroup
static inline int raster_faces_per_block(int F, int B = 1) {
    // few faces = big faces: fewer per workgroup.  With many images in the batch the machine is full anyway and
    // fatter workgroups (fewer rounds of workgroups) win.
    int r = 8;
    while (r < RF && (size_t)r * 256 < (size_t)F * B) r <<= 1;
    return r;
}
// raster workgroups per image (hand blocks, then object blocks): also the number of fragment-list segments per render
constexpr int RF_H_MIN = 2;  // fewest hand faces per raster workgroup a caller may ask for: sizes the fragment-list segments
static inline void raster_blocks(const foho_dims& d, int& rf_h, int& rf_o, int& nRh, int& nRo, int hand_faces = 0) {
    const int Fh_max = d.Fh_max > 0 ? d.Fh_max : d.Fmax, Fo_max = d.Fo_max > 0 ? d.Fo_max : d.Fmax;
    // hand faces are the big ones (tens of pixels each, the palm's up to hundreds): half of that per workgroup (a quarter overfills the chip at one image: a second round of workgroups), so that
    // the workgroup with the largest faces does not set the length of the launch
    rf_h = std::max(2, raster_faces_per_block(Fh_max, d.B) / 2);
    if (hand_faces > 0) rf_h = std::max(RF_H_MIN, std::min(hand_faces, RF));  // foho_step_desc.hand_faces_per_block
    rf_o = raster_faces_per_block(Fo_max, d.B);
    // development build: faces per raster workgroup from the environment (sweeps; foho_stamps.h)
    if (const char* e = FOHO_DEV_ENV("FOHO_DEBUG_RFH")) rf_h = std::max(1, std::min(atoi(e), RF));
    if (const char* e = FOHO_DEV_ENV("FOHO_DEBUG_RFO")) rf_o = std::max(1, std::min(atoi(e), RF));
    nRh = cdiv(Fh_max, rf_h);
    nRo = cdiv(Fo_max, rf_o);
}

static WS make_ws(const foho_dims& d) {
    WS w;
    size_t o = 0;
    const size_t P = (size_t)d.H * d.W, R = d.n_renders, B = d.B;
    const size_t V3 = (size_t)d.Vtot * 3 * 4;
    w.btiles_x = (d.W + BTX - 1) / BTX;
    w.nbtiles = w.btiles_x * ((d.H + BTY - 1) / BTY);
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = al(o + bytes);
        return r;
    };
    const int G1 = d.grid_res + 1;
    // --- static with the targets (FOHO_STAGE_TARGETS); first, so that their place does not move with the mesh sizes ---
    w.tile_static = take(B * (size_t)w.nbtiles * 12 * 4);  // loss contributions of the target maps per tile (k_loss.inc)
    w.image_static = take(B * 12 * 8);                     // ... and per image (double)
    // --- zeroed every step (atomic accumulators) ---
    w.zero_begin = o;
    w.frac_count = take(R * B * 4);
    {
        int rf_h, rf_o, nRh, nRo;
#ifdef FOHO_NSEG_AUTO
        raster_blocks(d, rf_h, rf_o, nRh, nRo, 0);
#else
        raster_blocks(d, rf_h, rf_o, nRh, nRo, RF_H_MIN);  // room for the most workgroups any hand_faces_per_block gives
#endif
        w.nseg = nRh + nRo;
    }
    w.seg_count = take(R * B * (size_t)w.nseg * 4);  // entries in every raster workgroup's segment of the fragment list
    w.kfix_count = take(R * B * 4);  // pixels whose K = 100 buffer was re-built this step (k_resolve -> frac_bwd_role)
    w.hit_count = take(R * B * 4);  // tiles with at least one hit pixel, per (render, image)
    w.bwd_count = take(R * B * 4);  // entries of the backward pass's work list (a dense tile is split into up to 4 entries)
    w.rstats = take(R * B * sizeof(RStats));
    w.rslot = take(R * B * NSLOT * sizeof(RSlot));
    w.g_world = take(V3);
    w.g_ndc = take(V3);  // dL/d(vertex NDC position), accumulated by k_pix_bwd over all renders
    w.g_nrm = take(V3);  // dL/d(unit vertex normal) = sum of the colour gradients of the incident faces
    w.parity = take(B * 2 * (size_t)G1 * G1 * 16);
    w.int_count = take(B * 4);
    w.final_ticket = take(B * 4);
    w.loss_acc = take(R * B * LOSS_SLOTS * NPART * 8);  // 12 loss sums per render x LOSS_SLOTS (double atomics of k_loss, read by k_pix_bwd)
    w.knn_inv = take(B * (size_t)std::max(d.Vh_max, 1) * 8);  // ~(d2 bits << 32 | index) of the nearest object vertex, atomicMax
    w.zero_end = o;
    // --- scatter planes of the rasteriser: all-zero outside [k_stage2, k_resolve]; k_resolve puts back to zero what
    // it consumed, FOHO_STAGE_BBOX clears everything (first use / after an aborted step)
    w.clean_begin = o;
    w.zkey = take(R * B * P * 8);  // ~(z bits << 32 | face id), atomicMax; 0 = no fragment
    w.fullb = take(R * B * P);     // 1 = some fragment covers the pixel fully (1 - p == 0): plain byte stores, every writer stores 1
    w.nfrac = take(R * B * P * 4); // fractional-coverage fragments per pixel (the K = 100 test); only they touch it
    w.plog = take(R * B * P * 8);  // sum of -log2(1 - p) over the fractional fragments in 2^-40 fixed point (integer atomics:
                                   // exact, order independent) -> their product; non-zero <=> the pixel has such a fragment
    w.tile_touched = take(R * B * (size_t)w.nbtiles);  // 1 = a face's pixel box overlaps the tile this step (raster setup)
    w.tile_clean = take(R * B * (size_t)w.nbtiles);    // 1 = the tile's p2f entries are known to be all -1
    w.sim_acc = take(2 * B * (size_t)SIM_ACC * 4);     // deferred update: per step parity, the 36 partial sums of the final stage (float atomics)
    w.pending = take(B * 4);                           // 1 = a deferred update of image b waits for the next k_xform / finalize
    w.clean_end = o;
    // --- plain scratch ---
    w.mesh_info = take(B * 2 * sizeof(MeshInfo));
    w.xf_part = take(B * 2 * VERT_BLOCKS_MAX * 8 * 4);
    w.vbox = take(B * (size_t)VERT_BLOCKS_MAX * 4 * 8 * 4);  // world-space AABB of every 64 consecutive object vertices (k_xform -> role_knn)
    w.world = take(V3);
    w.ndc = take(V3);
    w.vn_raw = take((size_t)d.Vtot * 16);  // per vertex: normalised raw normal (by the reciprocal) + the reciprocal norm
    w.vn = take(V3);
    w.face_ndc = take((size_t)d.Ftot * 9 * 4);
    w.state_next = take(B * (size_t)STATE_NEXT * 4);   // deferred update: params 16 | adam m 16 | adam v 16 | t | flags, written by k_xform
    w.pair_v = take((size_t)d.Ftot * 3 * 8);  // per (vertex, incident face) pair, CSR order: the face's 3 vertex ids + the corner (pack_pair)
    w.p2f = take(R * B * P * 4);
    w.zbuf = take(R * B * P * 4);
    w.sdist = take(R * B * P * 4);
    w.prod = take(R * B * P * 4);
    w.hit_list = take(R * B * (size_t)w.nbtiles * 4);  // ids of those tiles (k_resolve appends, k_loss walks the list)
    w.bwd_list = take(R * B * (size_t)w.nbtiles * BWD_SPLIT * 4);  // tile | part << 16 | parts << 20 (k_resolve appends, k_pix_bwd walks it)
    w.pcol = take(R * B * P * 12);  // colour n_a + n_b + n_c of the hit face (read back by the loss / backward passes)
    w.frac = take(R * B * (size_t)d.frac_cap * sizeof(FracEntry));  // overflow of the segments (rare)
    w.frac_seg = take(R * B * (size_t)w.nseg * FRAC_SEG * sizeof(FracEntry));
    w.act_list = take(R * B * (size_t)w.nbtiles * 4);  // batches: tiles k_resolve has work on this step (k_tile_list), per (render, image)
    w.act_count = take(R * B * 4);
    w.kfix = take(R * B * KFIX_MAX * sizeof(KFixEntry));
    w.loss_part = take(R * B * LOSS_BLOCKS * NPART * 4);
    w.stats2 = take(R * B * NSTAT * 4);
    w.g_direct = take(V3);
    w.knn_idx = take((size_t)d.Vtot * 4);
    w.knn_d2 = take((size_t)d.Vtot * 4);
    w.hand_order = take(B * (size_t)std::max(d.Vh_max, 1) * 4);  // lane slot -> hand vertex of the nearest-neighbour role, as a DELTA (all-zero = identity)
    w.kp3d = take(B * 21 * 3 * 4);
    w.g_kp3d = take(B * 21 * 3 * 4);
    w.vert_part = take(B * VERT_BLOCKS_MAX * 8 * 4);
    w.g_special = take(B * 2 * 6 * 4 * 4);  // moge-space gradient of the 6 arg-min / arg-max vertices of each mesh
    w.sim_part = take(B * 2 * (size_t)SIM_ROWS_MAX * SIM_NP * 4);
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
        case FOHO_WS_FRAG_COUNT: off = w.nfrac; n = R * B * P * 4; break;
        case FOHO_WS_SEG_COUNT: off = w.seg_count; n = R * B * (size_t)w.nseg * 4; break;
        case FOHO_WS_HAND_ORDER: off = w.hand_order; n = B * (size_t)std::max(d.Vh_max, 1) * 4; break;
        default: return -1;
    }
    if (nbytes) *nbytes = (int64_t)n;
    return (int64_t)off;
}

// One (vertex, incident face) pair in 8 bytes: v0 in 22 bits, v1 - v0 and v2 - v0 biased by 2^19 in 20 bits each (a face's
// vertices belong to one image, and an image holds fewer than 2^19 vertices: |difference| < 524288), the corner in the top
// 2 bits.  Needs Vtot <= 2^22 and Vmax < 2^19 (both checked on the host: pair_limits_ok); the fields are masked, so an id
// outside those limits can never spill into the neighbouring field.
__device__ __forceinline__ unsigned long long pack_pair(int v0, int v1, int v2, int corner) {
    return (unsigned long long)((unsigned)v0 & 0x3FFFFFu) | ((unsigned long long)((unsigned)(v1 - v0 + (1 << 19)) & 0xFFFFFu) << 22) |
           ((unsigned long long)((unsigned)(v2 - v0 + (1 << 19)) & 0xFFFFFu) << 42) | ((unsigned long long)(corner & 3) << 62);
}
__device__ __forceinline__ void unpack_pair(unsigned long long p, int v[3], int& corner) {
    v[0] = (int)(p & 0x3FFFFFu);
    v[1] = v[0] + (int)((p >> 22) & 0xFFFFFu) - (1 << 19);
    v[2] = v[0] + (int)((p >> 42) & 0xFFFFFu) - (1 << 19);
    corner = (int)(p >> 62);
}

// kernel-side view of the step (device pointers, by value)
struct Ctx {
    foho_dims d;
    PixAxis ax, ay;  // pixel column / row -> NDC (pix_to_ndc)
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
    int32_t* p2f;
    float *zbuf, *sdist, *prod, *pcol;
    int* hit_list;
    unsigned* hit_count;
    int* bwd_list;
    unsigned* bwd_count;
    uint8_t *tile_touched, *tile_clean;
    int* act_list;        // k_tile_list -> k_resolve (batches)
    unsigned* act_count;
    unsigned long long* pair_v;
    int* pending;
    float* state_next;
    float* sim_acc;
    unsigned long long* zkey;
    uint8_t* fullb;
    unsigned* nfrac;
    unsigned long long* plog;
    FracEntry* frac;
    unsigned* frac_count;
    FracEntry* frac_seg;
    unsigned* seg_count;
    unsigned* kfix_count;
    KFixEntry* kfix;
    int nseg;
    RStats* rstats;
    RSlot* rslot;
    float *loss_part, *stats2;
    float *g_ndc, *g_nrm, *g_world, *g_direct;
    int32_t* knn_idx;
    const int32_t* hand_order;
    float *knn_d2, *kp3d, *g_kp3d, *vert_part, *sim_part, *xf_part, *g_special, *vbox;
    unsigned long long* parity;
    int32_t* int_count;
    unsigned* final_ticket;
    unsigned long long* knn_inv;
    double* loss_acc;
    float* tile_static;
    double* image_static;
    int btiles_x;
};

// render r of a by-value kernel argument: a run-time index would make the compiler copy the array to scratch memory
__device__ __forceinline__ foho_render_cfg pick_render(const foho_render_cfg (&rr)[2], int r) {
    foho_render_cfg o = rr[0];
    if (r != 0) o = rr[1];
    return o;
}

// ---- G-buffer planes of the hit pixels: depth and face colour, fp32 or (dims.gbuf_f16, BASELINE configs[4]) fp16 ----
// Selection (face ids, the z keys) and every sum stay fp32; with gbuf_f16 the stored depth and colour are rounded to half
// precision ONCE, in k_resolve, before the render's extrema are taken, so that the loss and backward passes -- which
// convert back to fp32 -- see exactly the values the extrema were computed from (their `== max` tie tests stay exact).
__device__ __forceinline__ float round_f16(float x) { return __half2float(__float2half(x)); }
__device__ __forceinline__ void gbuf_store(const Ctx& c, size_t pi, float z, const float* col) {
    if (c.d.gbuf_f16) {
        __half* zh = reinterpret_cast<__half*>(c.zbuf);
        __half* ch = reinterpret_cast<__half*>(c.pcol);
        zh[pi] = __float2half(z);
        if (col)
            for (int k = 0; k < 3; k++) ch[3 * pi + k] = __float2half(col[k]);
    } else {
        c.zbuf[pi] = z;
        if (col)
            for (int k = 0; k < 3; k++) c.pcol[3 * pi + k] = col[k];
    }
}
__device__ __forceinline__ void gbuf_load(const Ctx& c, size_t pi, float& z, float* col) {
    if (c.d.gbuf_f16) {
        const __half* zh = reinterpret_cast<const __half*>(c.zbuf);
        const __half* ch = reinterpret_cast<const __half*>(c.pcol);
        z = __half2float(zh[pi]);
        for (int k = 0; k < 3; k++) col[k] = __half2float(ch[3 * pi + k]);
    } else {
        z = c.zbuf[pi];
        for (int k = 0; k < 3; k++) col[k] = c.pcol[3 * pi + k];
    }
}

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

__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ __forceinline__ void mesh_center(const MeshInfo& mi, float* cen) {
    for (int k = 0; k < 3; k++) {
        const float mn = ord2f((unsigned)((~mi.kmin_inv[k]) >> 32)), mx = ord2f((unsigned)(mi.kmax[k] >> 32));
        cen[k] = (mn + mx) / 2.0f;
    }
}
__device__ __forceinline__ int mesh_argmin(const MeshInfo& mi, int k) { return (int)(unsigned)((~mi.kmin_inv[k]) & 0xffffffffull); }
__device__ __forceinline__ int mesh_argmax(const MeshInfo& mi, int k) { return (int)(~(unsigned)(mi.kmax[k] & 0xffffffffull)); }

#include "k_vertex.inc"
#include "k_inside.inc"
#include "k_raster.inc"
#include "k_loss.inc"
#include "k_final.inc"
#include "k_xform.inc"
#include "k_backward.inc"
#include "host.inc"
#include "ops.inc"
#include "k_sdf.inc"
#include "k_lbs.inc"
#include "k_icp.inc"
#include "k_flexi.inc"
#include "k_topo.inc"
#include "k_object.inc"
#include "mesh_decimate.inc"
