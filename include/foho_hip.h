/*
 * foho_hip.h -- C ABI of libfoho_hip.so, the MI355X (gfx950) implementation of FollowMyHold's
 * guidance / alignment hot path.
 *
 * Boundary rules:
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *   - The caller owns every buffer (PyTorch-ROCm allocates them); the library allocates no
 *     persistent device memory.  Scratch comes from a caller-provided workspace whose size is
 *     queried with foho_step_workspace_bytes().
 *   - Every launch is asynchronous on the hipStream_t passed as `void* stream`
 *     (torch.cuda.current_stream().cuda_stream); no implicit device synchronisation.
 *   - Return 0 on success, a negative foho_status otherwise; foho_last_error() gives a
 *     thread-local message.  Nothing throws across the ABI.
 *
 * Reference interfaces replaced (paths relative to the reference repo;
 * PL = third_party_patches/hy3dgen/shapegen/pipelines.py, RUN = src/foho/guidance/run.py,
 * SDF = third_party/utilz/kaolin_sdf_ops.py, ICP = src/foho/alignment/mesh_align.py):
 *
 *   foho_step_run            one optimisation iteration of PL:1320-1358 (phase A),
 *                            PL:1386-1453 (phase B) or PL:1480-1601 (phase C, the "guidance step"):
 *                            pytorch3d MeshRasterizer/MeshRenderer (RUN:95-116), PhongNormalShader
 *                            (PL:74-92), render_normal_and_disparity (PL:272-289), loss heads
 *                            (PL:178-186, 1338-1349, 1421-1440, 1495-1504, 1539-1541, 1567-1588),
 *                            knn_points (PL:1529-1532), mesh_edge_loss (PL:1575),
 *                            kaolin_sdf.get_sdf_of_meshes + honerf_intersection_loss
 *                            (PL:1553-1554, SDF:131-160, PL:231-239), loss.backward() and
 *                            torch.optim.Adam/AdamW(eps=1e-4).step() (PL:1357-1358, 1452-1453, 1600-1601)
 *   foho_raster_fwd/_bwd     pytorch3d rasterize_meshes fwd/bwd as called by MeshRasterizer (PL:273-274)
 *   foho_knn1_fwd            pytorch3d.ops.knn_points(K=1) (PL:1529-1538)
 *   foho_inside_points       kaolin.ops.mesh.check_sign (SDF:104); the joint 65^3 grid variant is fused in foho_step_run
 *   foho_point_mesh_dist     kaolin.metrics.trianglemesh.point_to_mesh_distance (SDF:101)
 *   foho_lbs_fwd/_bwd        smplx MANOLayer forward (third_party/estimator/hamer/hamer/models/hamer.py:125-130)
 *   foho_icp_run             icp() loop: cKDTree.query + trimmed procrustes + scale clip (ICP:104-142)
 *   foho_icp_run_batch       the same for all start transforms of icp() at once (ICP:91-175)
 *   foho_icp_run_surface     the same with on_surface=True: closest point on the target triangles (ICP:106-107)
 *   foho_mesh_decimate       hy3dgen FaceReducer = pymeshlab quadric edge collapse (RUN:163), host code
 *   foho_geo_decode_fwd      the chunked `vae.geo_decoder(queries, latents)` loop of latent2sdf (PL:298-308)
 *   foho_vae_fwd/_bwd        `pred = vae(pred)` of latent2sdf (PL:295): the ShapeVAE transformer and its backward to the tokens
 *   foho_sdpa_fwd/_bwd       torch.nn.functional.scaled_dot_product_attention inside that transformer (fallback route)
 */
#ifndef FOHO_HIP_H
#define FOHO_HIP_H

#include <stddef.h>
#include <stdint.h>

/* Every entry point below is FOHO_API; the library is built with -fvisibility=hidden, so these (and nothing else: no kernel
 * stubs, no helpers) are what `nm -D libfoho_hip.so` shows. */
#ifndef FOHO_API
#define FOHO_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FOHO_OK = 0,
    FOHO_ERR_BAD_ARG = -1,
    FOHO_ERR_LAUNCH = -2,
    FOHO_ERR_WORKSPACE = -3
} foho_status;

FOHO_API const char* foho_last_error(void);
FOHO_API int foho_version(void);
/* Layout check for bindings that mirror the structs (ctypes, cgo, JNI): fills out[0..4] with sizeof(foho_image),
 * sizeof(foho_dims), sizeof(foho_render_cfg), sizeof(foho_step_cfg), sizeof(foho_step_desc) of THIS build and returns
 * foho_version().  A caller whose own sizes differ is talking to another version of the ABI (fields are only ever
 * appended; entry points that gained parameters: foho_raster_bwd's blur_radius in 101). */
FOHO_API int foho_abi_sizes(int64_t out[5]);

/* ---- per-image constants (array of B of these lives in DEVICE memory) -------------------- */
typedef struct {
    int32_t v_off, Vh, Vo;   /* scene vertices [v_off, v_off+Vh+Vo): hand first, then object   */
    int32_t f_off, Fh, Fo;   /* scene faces    [f_off, f_off+Fh+Fo): hand first (PL:1544)      */
    int32_t n_edges;         /* unique undirected edges of the object mesh (mesh_edge_loss)    */
    int32_t jcols;           /* columns of the joint regressor (778)                           */
    float k00, k11;          /* FoVPerspectiveCameras K[0,0], K[1,1] (RUN:90)                   */
    float cam_R[9];          /* row-vector convention: X_view = X_world @ R + T                */
    float cam_T[3];
    float znear, zfar;
    float T_h2m[12];         /* rows of the 3x4 Hunyuan->MoGe transform (PL:1240-1241)          */
} foho_image;

/* ---- sizes (host) -------------------------------------------------------------------------- */
typedef struct {
    int32_t B, H, W;
    int32_t Vtot, Ftot;      /* packed totals over the batch                                     */
    int32_t Vmax, Fmax;      /* per-image maxima of Vh+Vo and Fh+Fo (grid sizing)                */
    int32_t Vh_max, Vo_max;  /* per-image maxima of the hand / object vertex counts              */
    int32_t grid_res;        /* SDF grid resolution (64 -> 65^3 points; PL:1126, SDF:146)        */
    int32_t frac_cap;        /* capacity of the fractional-coverage fragment list per render     */
    int32_t n_renders;       /* 1 (phase A/B) or 2 (phase C)                                     */
    int32_t Fh_max, Fo_max;  /* largest per-image hand / object face count (0 = unknown: Fmax is used) */
    int32_t gbuf_f16;        /* 1: depth and colour planes of the G-buffer in fp16 (BASELINE configs[4]: "fp16 rasterizer +
                                fp32 loss accumulate"); face selection, edge distances and all sums stay fp32        */
} foho_dims;

/* ---- one render = one mesh through renderer + sil_renderer --------------------------------- */
enum { FOHO_FACES_HAND = 0, FOHO_FACES_OBJ = 1, FOHO_FACES_ALL = 2 };
enum { FOHO_MASK_NONE = 0, FOHO_MASK_HAND = 1, FOHO_MASK_OBJ = 2, FOHO_MASK_HOI = 3 };
typedef struct {
    int32_t face_set;        /* FOHO_FACES_*                                                      */
    int32_t normal_mask;     /* valid_mask of normal_alignment_loss (FOHO_MASK_*)                 */
    int32_t disp_mask;       /* mask multiplied into the disparity TARGET (PL:1340; NONE at PL:1568) */
    int32_t sil_mask;        /* BCE target mask                                                   */
    float w_normal, w_disp, w_sil;   /* effective weights inside the total loss                   */
} foho_render_cfg;

/* ---- loss weights / optimiser (host, by value) ---------------------------------------------- */
typedef struct {
    foho_render_cfg render[2];
    float w_kps, w_trans_hand, w_trans_obj, w_verts_obj, w_edge, w_contact;
    float contact_margin;            /* 0.01 on the squared distance (PL:1539)                    */
    int32_t use_intersection;        /* CFG use_intersection_loss                                 */
    float w_int_near, w_int_far;     /* 1e-5 / 1e-9 (PL:1561-1564)                                 */
    float int_gate;                  /* mean(d^2) < 0.001 (PL:1561)                                */
    int32_t int_gate_step_ok;        /* i >= num_inference_steps - 3, evaluated by the host        */
    float sigma, gamma, blur_radius; /* BlendParams / RasterizationSettings (RUN:91-101)           */
    /* Adam / AdamW over the 16 similarity parameters [s_h t_h(3) q_h(4) s_o t_o(3) q_o(4)]        */
    float lr[16];                    /* 0 = frozen (phase A freezes the object, phase B the hand) */
    float beta1, beta2, eps, weight_decay;
    int32_t do_update;               /* 0: gradients only                                          */
    int32_t world_space_input;       /* 1: the similarity transform is skipped (vertices are used as they are, after
                                        T_h2m for the object) -- rendering target maps of a fixed mesh (PL:1247-1256),
                                        where the reference does not apply transform_mesh_around_center either      */
    int32_t deferred_update;         /* 1 + (k & 1) for the k-th step of a sequence, 0 = off.  The step ends after k_vert_bwd
                                        has accumulated its partial sums (double buffered by that parity); loss assembly,
                                        parameter gradients and the optimiser update are applied by the prologue of the
                                        NEXT step's first kernel (every workgroup recomputes them, one stores them) --
                                        for back-to-back steps inside one hipGraph, where it removes the serial tail of
                                        the last-workgroup stage.  params / adam / losses / flags of the last step only
                                        become visible after foho_step_finalize()                                    */
    int32_t n_active_renders;        /* 0 = all dims.n_renders.  1 with a workspace laid out for 2 renders: phases A and B
                                        (one render each, PL:1328-1329, 1414-1415) run on the workspace phase C uses, so a
                                        per-image job never re-allocates or re-initialises it between phases.  Roles whose
                                        output has no weight in this cfg are not launched at all: a mesh no active render
                                        draws gets no raster / vertex-normal workgroups, w_contact == 0 without the
                                        intersection gate -> no nearest-neighbour role (contact / mean_d2 report 0),
                                        w_edge == w_verts_obj == 0 -> no edge role, w_kps == 0 -> no keypoint role        */
    int32_t listed_cap;              /* how the resolve pass finds its tiles.  0 = automatic: one workgroup per 32x8 tile of
                                        the frame below eight images per launch, a compacted list of the tiles that need
                                        work (k_tile_list) walked by 3/8 of the frame's tile count as workgroups from eight
                                        on; < 0 = always the former; n > 0 = always the latter with n workgroups per
                                        (render, image).  Same results either way.  (Was the FOHO_LISTED_CAP environment
                                        variable up to version 101; the library reads no environment any more.)        */
} foho_step_cfg;

/* ---- buffers of one batched step -------------------------------------------------------------- */
typedef struct {
    foho_dims dims;
    const foho_image* images;        /* device, B entries                                          */
    /* geometry (device) */
    const float* verts_in;           /* (Vtot,3): hand part = mano_mesh_moge, object part = Hunyuan space */
    const int32_t* faces;            /* (Ftot,3) GLOBAL vertex ids                                  */
    const int32_t* inc_off;          /* (Vtot+1) CSR vertex -> incident (face<<2 | corner)          */
    const int32_t* inc_fc;           /* (3*Ftot)                                                    */
    const int32_t* nbr_off;          /* (Vtot+1) CSR vertex -> neighbour vertices over unique edges */
    const int32_t* nbr_idx;          /* (2*E)                                                        */
    const float* J_regressor;        /* (16, jcols) shared by the batch                             */
    /* targets (device) */
    const float* tgt_normal;         /* (B,H,W,3)  moge_normal (PL:1253)                             */
    const float* tgt_disp;           /* (B,H,W)    moge_disp   (PL:1254)                             */
    const uint8_t* mask;             /* (B,H,W)    bit0 = hand mask, bit1 = object mask              */
    const float* kps_2d;             /* (B,21,2)   hamer_2d_kps (PL:1220)                            */
    /* state (device, updated in place) */
    float* params;                   /* (B,16)                                                       */
    float* adam_m;                   /* (B,16)                                                       */
    float* adam_v;                   /* (B,16)                                                       */
    int32_t* adam_t;                 /* (B)   step counters; also hold the sticky NaN-break flag     */
    /* outputs (device) */
    float* losses;                   /* (B,FOHO_N_LOSS)                                              */
    float* grad_params;              /* (B,16)                                                       */
    float* grad_verts_in;            /* (Vtot,3)  dL/d verts_in (object rows are the autograd sink)  */
    int32_t* flags;                  /* (B)  bit0 NaN loss, bit1 frac list overflow, bit2 >K faces/pixel,
                                        (bit3 retired in 101: near-plane clipping is implemented),
                                        bit4 / bit5: capacity mode, see foho_object_update                      */
    void* workspace;
    size_t workspace_bytes;
    int32_t hand_order_valid;        /* 1: the caller has written a permutation into FOHO_WS_HAND_ORDER (below); 0: the
                                        region is ignored and lanes take hand vertices in index order -- a workspace that
                                        is not zero-filled (plain hipMalloc) is safe with 0                          */
    int32_t hand_faces_per_block;    /* hand faces per workgroup of the scatter rasteriser: 0 = automatic (4 at one image per
                                        launch, more in batches); 2 .. 64 = the caller's choice.  The library cannot see how
                                        large the hand is on screen (images[] lives in device memory): on the reference's
                                        crops (meshes fill the frame, fov ~ 25 degrees) a hand face covers 7 x the pixels it
                                        does through a 60 degree lens, and 2 per workgroup is 7 % faster at one image per
                                        launch (4 is 3 % faster on the wide frames).  Same results either way.       */
} foho_step_desc;

/* indices into losses[b][*] */
enum {
    FOHO_L_TOTAL = 0, FOHO_L_INTERSECTION, FOHO_L_CONTACT, FOHO_L_KPS, FOHO_L_TRANS_HAND, FOHO_L_TRANS_OBJ,
    FOHO_L_VERTS_OBJ, FOHO_L_EDGE, FOHO_L_NORMAL0, FOHO_L_DISP0, FOHO_L_SIL0, FOHO_L_NORMAL1, FOHO_L_DISP1,
    FOHO_L_SIL1, FOHO_L_N_INTERSECT, FOHO_L_W_INT, FOHO_L_MEAN_D2, FOHO_N_LOSS = 24
};

/* stage mask of foho_step_run (tests and profiling run prefixes of the step) */
enum {
    FOHO_STAGE_VERTEX = 1, FOHO_STAGE_RASTER = 2, FOHO_STAGE_LOSS = 4, FOHO_STAGE_BACKWARD = 8,
    FOHO_STAGE_INSIDE = 16, FOHO_STAGE_FINAL = 32,
    FOHO_STAGE_BBOX = 64,            /* per-input state in the workspace: AABB of verts_in (centre of the similarity
                                        transform, PL:111), clean rasteriser planes, the (vertex, face) pair table of
                                        the topology: needed once per workspace and again whenever the caller
                                        rewrites verts_in, faces or the incidence tables                         */
    FOHO_STAGE_STEP = 63,            /* one optimisation step with a cached AABB                               */
    FOHO_STAGE_ALL = 127,
    FOHO_STAGE_TARGETS = 256         /* per-tile / per-image sums of the TARGET maps (tgt_disp, mask) in the workspace: the
                                        loss pass only visits tiles with a hit and takes the rest from these.  Needed once
                                        per workspace before the first FOHO_STAGE_LOSS, and again when the caller rewrites
                                        tgt_normal / tgt_disp / mask                                               */
};

/* named workspace regions, for parity tests that inspect intermediates.
 * FOHO_WS_KNN_IDX persists between steps: the nearest-neighbour role prunes with the distance to the vertex it names (any
 * content is valid, the true previous answer prunes best).
 * FOHO_WS_HAND_ORDER (B x Vh_max int32, host-written, optional, read only when foho_step_desc.hand_order_valid): which hand
 * vertex lane `slot` of the nearest-neighbour role takes, stored as a DELTA on the slot -- all-zero is the identity.  A spatially
 * coherent order (the Python host writes the Morton order of the input hand) lets the 64 lanes of a wave skip the same runs
 * of candidates; any PERMUTATION of [0, Vh) gives identical results.  Rewrite it after the workspace is re-allocated. */
enum {
    FOHO_WS_WORLD = 0, FOHO_WS_NDC, FOHO_WS_VN, FOHO_WS_P2F, FOHO_WS_ZBUF, FOHO_WS_SDIST, FOHO_WS_PROD,
    FOHO_WS_KNN_IDX, FOHO_WS_KNN_D2, FOHO_WS_GWORLD, FOHO_WS_FRAC_COUNT, FOHO_WS_STATS, FOHO_WS_PARITY,
    FOHO_WS_FRAG_COUNT, FOHO_WS_SEG_COUNT, FOHO_WS_HAND_ORDER, FOHO_WS_NREGIONS
};

FOHO_API size_t foho_step_workspace_bytes(const foho_dims* dims);
/* byte offset and byte length of a named region inside the workspace (-1 on bad id) */
FOHO_API int64_t foho_step_workspace_region(const foho_dims* dims, int region, int64_t* nbytes);
/* One step (or the prefix of it that stage_mask names) on `stream`, asynchronously.  Everything that shapes the launches is in
 * desc / cfg (cfg->listed_cap: how the resolve pass finds its tiles); the process environment is not consulted. */
FOHO_API int foho_step_run(const foho_step_desc* desc, const foho_step_cfg* cfg, int stage_mask, void* stream);
/* Applies the update a deferred_update step left pending (no-op per image when nothing is pending): one small launch;
 * cfg->deferred_update must carry the number of the LAST step run. */
FOHO_API int foho_step_finalize(const foho_step_desc* desc, const foho_step_cfg* cfg, void* stream);
/* Runs foho_step_run(FOHO_STAGE_STEP) TWICE: an un-timed iteration (with the other parity when cfg->deferred_update is
 * set), then the same iteration again with every launch bracketed by hipEvents on `stream`; synchronises the stream and
 * returns the duration of each launch of the second iteration in milliseconds (measurement aid for bench.py);
 * foho_kernel_name(i) names the i-th launch of the calling thread's last profiled run ("" past the end). */
#define FOHO_N_KERNELS 10
FOHO_API int foho_step_run_profiled(const foho_step_desc* desc, const foho_step_cfg* cfg, void* stream, float* ms_out);
FOHO_API const char* foho_kernel_name(int i);

/* ---- stand-alone operators (facade level) -------------------------------------------------- */
/* pytorch3d rasterize_meshes(faces_per_pixel=1) on NDC vertices of ONE mesh.
 * verts_ndc (V,3) = (x_ndc, y_ndc, z_view); faces (F,3) int32.  Outputs (H,W): pix_to_face int64
 * (-1 background), zbuf, bary (H,W,3), dists (signed, squared NDC).  sil_prod (H,W) optional:
 * prod_k(1 - sigmoid(-d_k/sigma)) over every fragment of the pixel (SoftSilhouetteShader alpha = 1 - it).
 * Near plane (MeshRasterizer's z_clip_value = znear / 2 = 0.005 for the path's camera, RUN:84-105): faces entirely nearer
 * are culled, faces that straddle it are rasterised as the one or two sub-triangles pytorch3d's clip_faces cuts them into
 * (pix_to_face and bary refer to the unclipped face); *overflow_flag gets bit2 (4) when the K = 100 cut-off could not be
 * reproduced. */
FOHO_API int foho_raster_fwd(const float* verts_ndc, const int32_t* faces, int32_t V, int32_t F, int32_t H, int32_t W,
                    float blur_radius, float sigma, int64_t* pix_to_face, float* zbuf, float* bary, float* dists,
                    float* sil_prod, int32_t* overflow_flag, void* workspace, size_t workspace_bytes, void* stream);
FOHO_API size_t foho_raster_workspace_bytes(int32_t V, int32_t F, int32_t H, int32_t W);
/* backward of the K=1 fragments: grad_verts_ndc (V,3) += d(zbuf,bary,dists)/d verts_ndc.  blur_radius = the forward
 * call's (it decides which half of a near-clipped face left a fragment; for such faces grad_bary is taken w.r.t. the
 * sub-triangle's barycentrics) */
FOHO_API int foho_raster_bwd(const float* verts_ndc, const int32_t* faces, int32_t V, int32_t F, int32_t H, int32_t W,
                    const int64_t* pix_to_face, const float* grad_zbuf, const float* grad_bary,
                    const float* grad_dists, float* grad_verts_ndc, float blur_radius, void* stream);
/* backward of sil_prod (version 102): grad_verts_ndc (V,3) += d(sil_prod)/d verts_ndc . grad_prod, i.e. the gradient of
 * SoftSilhouetteShader's alpha = 1 - sil_prod through every fragment of the pixels with 0 < sil_prod < 1 (the only ones that
 * carry one: d prod / d sdist_k = prod sigmoid(-sdist_k / sigma) / sigma).  sil_prod: foho_raster_fwd's output; blur_radius and
 * sigma: that call's.  The product over ALL fragments is differentiated -- identical to the K = 100 product unless a pixel
 * holds 100 fractional-coverage fragments or more. */
FOHO_API int foho_raster_sil_bwd(const float* verts_ndc, const int32_t* faces, int32_t V, int32_t F, int32_t H, int32_t W,
                        const float* sil_prod, const float* grad_prod, float* grad_verts_ndc, float blur_radius, float sigma,
                        void* stream);
/* K=1 nearest neighbour: d2 (N1), idx (N1) int64; ties -> lowest index */
FOHO_API int foho_knn1_fwd(const float* p1, int32_t N1, const float* p2, int32_t N2, float* d2, int64_t* idx, void* stream);

/* kaolin.metrics.trianglemesh.point_to_mesh_distance (SDF:101): exact squared distance of N points to the
 * closest triangle of ONE mesh and that triangle's index (ties -> lowest index); face_idx may be NULL. */
FOHO_API int foho_point_mesh_dist(const float* verts, const int32_t* faces, int32_t V, int32_t F, const float* pts, int32_t N,
                         float* d2, int64_t* face_idx, void* stream);
/* kaolin.ops.mesh.check_sign (SDF:104): inside[n] = 1 when pts[n] is inside the closed mesh (+z ray parity,
 * rays through edges / vertices counted once). */
FOHO_API int foho_inside_points(const float* verts, const int32_t* faces, int32_t V, int32_t F, const float* pts, int32_t N,
                       uint8_t* inside, void* stream);

/* smplx MANOLayer(pose2rot=False) forward (hamer/models/hamer.py:125-130).  Model arrays (device,
 * float32): v_template (V,3), shapedirs (V,3,10), posedirs (135,3V), J_regressor (16,V), lbs_weights (V,16),
 * parents (16) int32.  betas (B,10), rot_mats (B,16,3,3) = [global_orient | hand_pose].  Outputs verts (B,V,3),
 * joints (B,16,3) posed joints (may be NULL).  use_mfma: 1 = pose-blend contraction on the f32 matrix cores,
 * 0 = per-vertex dot products, -1 = automatic (matrix cores when B >= 16). */
FOHO_API size_t foho_lbs_workspace_bytes(int32_t B, int32_t V);
FOHO_API int foho_lbs_fwd(const float* v_template, const float* shapedirs, const float* posedirs, const float* J_regressor,
                 const float* lbs_weights, const int32_t* parents, int32_t V, const float* betas, const float* rot_mats,
                 int32_t B, int32_t use_mfma, float* verts, float* joints, void* workspace, size_t workspace_bytes,
                 void* stream);
/* backward of the last foho_lbs_fwd run on the same workspace: grad_verts (B,V,3), grad_joints (B,16,3) or NULL
 * -> grad_betas (B,10), grad_rot_mats (B,16,3,3) */
FOHO_API int foho_lbs_bwd(const float* v_template, const float* shapedirs, const float* posedirs, const float* J_regressor,
                 const float* lbs_weights, const int32_t* parents, int32_t V, const float* rot_mats, int32_t B,
                 const float* grad_verts, const float* grad_joints, float* grad_betas, float* grad_rot_mats,
                 void* workspace, size_t workspace_bytes, void* stream);

/* icp() of src/foho/alignment/mesh_align.py:56-175 for one start transform on already-sampled point sets
 * (float64, device): n_iter iterations of nearest neighbour -> drop the n_outliers largest distances ->
 * trimesh-style procrustes (reflection=False, scale = !fixed_scale) -> T = next @ T -> scale clipped to
 * [min_scale, max_scale].  T_out (4x4 row-major) is the transform stored with the lowest mean inlier distance,
 * cost_out that distance, cost_history (n_iter) optional.  Asynchronous: 2 launches per iteration, no host sync. */
FOHO_API size_t foho_icp_workspace_bytes(int32_t N, int32_t M);
FOHO_API int foho_icp_run(const double* src, int32_t N, const double* tgt, int32_t M, int32_t n_iter, int32_t n_outliers,
                 int32_t fixed_scale, double min_scale, double max_scale, double* T_out, double* cost_out,
                 double* cost_history, void* workspace, size_t workspace_bytes, void* stream);

/* The multi-start loop of icp() (`for cube in cubes`, ICP:91-175: identity + 7 reflections + 9 axis rotations) as
 * ONE enqueue: src (n_starts, N, 3) holds the already transformed start point sets, the target is shared; every
 * iteration is 2 launches whose grids carry the start index, so the 17 coarse starts cost the launches of one.
 * T_out (n_starts, 16), cost_out (n_starts), cost_history (n_starts, n_iter) optional; per start identical to
 * foho_icp_run (which is this call with n_starts = 1). */
FOHO_API size_t foho_icp_batch_workspace_bytes(int32_t n_starts, int32_t N, int32_t M);
FOHO_API int foho_icp_run_batch(const double* src, int32_t n_starts, int32_t N, const double* tgt, int32_t M, int32_t n_iter,
                       int32_t n_outliers, int32_t fixed_scale, double min_scale, double max_scale, double* T_out,
                       double* cost_out, double* cost_history, void* workspace, size_t workspace_bytes, void* stream);

/* icp(..., on_surface=True) (ICP:106-107): q = trimesh.proximity.closest_point(target_mesh, p), the closest point ON the
 * target triangles (float64 brute force over all Ft triangles, Voronoi-region test per triangle) instead of the nearest
 * sampled target point; everything else as foho_icp_run_batch.  tgt_verts (Vt,3) float64, tgt_faces (Ft,3) int32. */
FOHO_API size_t foho_icp_surface_workspace_bytes(int32_t n_starts, int32_t N, int32_t Ft);
FOHO_API int foho_icp_run_surface(const double* src, int32_t n_starts, int32_t N, const double* tgt_verts, int32_t Vt,
                         const int32_t* tgt_faces, int32_t Ft, int32_t n_iter, int32_t n_outliers, int32_t fixed_scale,
                         double min_scale, double max_scale, double* T_out, double* cost_out, double* cost_history,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- iso-surfacing between the diffusion latent and the guidance path ----------------
 * kaolin FlexiCubes.__call__(x_nx3, s_n, cube_fx8, res) with default weights, as called at pipelines.py:1393 / 1509
 * = Dual Marching Cubes on the regular (res+1)^3 grid: x (G^3,3) grid positions and s (G^3) SDF (negative inside),
 * grid point (i,j,k) at (i*G + j)*G + k (generate_dense_grid_points, PL:341-360).  Outputs: verts (up to verts_cap x 3),
 * faces (up to faces_cap x 3, int64, outward orientation), l_dev per vertex (optional), counts (device int32[3]:
 * vertices, triangles, overflow bits -- bit0 vertex capacity, bit1 face capacity).  Vertex order: (cube, patch); face
 * order: (axis, i, j, k) of the sign-change grid edge.  The workspace keeps what foho_flexi_bwd needs. */
FOHO_API size_t foho_flexi_workspace_bytes(int32_t res);
FOHO_API int foho_flexi_fwd(const float* x, const float* s, int32_t res, float* verts, int32_t verts_cap, int64_t* faces,
                   int32_t faces_cap, float* l_dev, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);
/* grad_s (G^3) and optional grad_x (G^3,3) must be zeroed by the caller; accumulated with float atomics. */
FOHO_API int foho_flexi_bwd(const float* x, const float* s, int32_t res, const float* grad_verts, int32_t n_verts, float* grad_s,
                   float* grad_x, const void* workspace, size_t workspace_bytes, void* stream);

/* ---- topology tables of a packed mesh, built on the device (at set-up, and every time the object's connectivity
 * changes: the FlexiCubes output of an iteration, pipelines.py:1393 / 1509).  faces (F,3) with vertex ids in [0,V).
 * inc_off (V+1) / inc_fc (3F) = vertex -> incident (face << 2 | corner) in pytorch3d's index_add order
 * (verts_normals_packed, PL:83).  With obj_flag (V bytes, 1 = object vertex): nbr_idx (3F) = vertex -> neighbours over
 * the unique edges (edges_packed / mesh_edge_loss, PL:1575) at the SAME offsets inc_off, valid when the flagged vertices
 * form closed, consistently oriented 2-manifolds (then n_edges = 3 F_obj / 2); *flag != 0 afterwards means they do not
 * (bit1) or a valence exceeds 48 (bit0) and the caller must build the tables with a general sort.  obj_flag == NULL:
 * incidence lists only. */
FOHO_API size_t foho_topology_workspace_bytes(int32_t V);
FOHO_API int foho_topology_tables(const int32_t* faces, int32_t V, int32_t F, const uint8_t* obj_flag, int32_t* inc_off, int32_t* inc_fc,
                         int32_t* nbr_idx, int32_t* flag, void* workspace, size_t workspace_bytes, void* stream);

/* ---- capacity mode: a new object mesh (new vertex / face counts, new connectivity) every iteration without a host
 * round trip -- what phases B and C of the reference do 550 times per image (pipelines.py:1391-1393, 1507-1509: decode
 * the latent, extract the iso-surface, build a fresh Meshes object).  The caller sizes every buffer of the step for a
 * per-image CAPACITY: dims.Vo_max / dims.Fo_max object vertices / faces per image, images[b].v_off / f_off strided
 * accordingly, inc_off with Vtot + 1 entries and nbr_off == inc_off (neighbour lists share the incidence offsets), and
 * lets foho_flexi_fwd write image b's vertices straight into verts_in + 3 * (images[b].v_off + images[b].Vh).
 * foho_object_update then, reading the ACTUAL counts from device memory (`counts`: B x 3 int32 as foho_flexi_fwd leaves
 * them), (1) stores Vo / Fo / n_edges = 3 Fo / 2 in the device-resident images[b] -- every kernel of the step takes its
 * bounds from there --, (2) converts obj_faces (B, faces_cap, 3) int64 mesh-local ids into int32 global ids behind the
 * hand's faces, (3) rebuilds the topology tables (foho_topology_tables semantics) and the (vertex, face) pair table,
 * (4) recomputes the AABB of the input meshes (centre of the similarity transform, PL:111).  5 launches, no host sync,
 * fixed addresses: foho_flexi_fwd -> foho_object_update -> foho_step_run(FOHO_STAGE_STEP) -> foho_flexi_bwd can be
 * captured in one hipGraph.  flags[b] bit6 (64): the iso-surface is empty (the reference skips such an iteration,
 * PL:1511-1513), bit4 (16): capacity overflow (the object is left empty for that step), bit5 (32):
 * the object is not a closed, consistently oriented 2-manifold or a valence exceeds 48 (the edge tables assume it; the
 * caller must fall back to exact-size tables built by a general sort).  While any of these three bits is set the
 * step computes losses and gradients but leaves parameters and optimiser state alone; the caller clears the bits once
 * it has dealt with them.  obj_flag (Vtot bytes): 1 = object vertex slot. */
FOHO_API size_t foho_object_workspace_bytes(int32_t Vtot, int32_t Ftot);   /* zero-fill it before the first use */
FOHO_API int foho_object_update(const foho_step_desc* desc, const int32_t* counts, const int64_t* obj_faces, int32_t faces_cap,
                       const uint8_t* obj_flag, void* workspace, size_t workspace_bytes, void* stream);

/* ---- mesh post-processing after the pipeline ------------------------------------------
 * `obj_mesh = FaceReducer()(obj_mesh)` (src/foho/guidance/run.py:163): hy3dgen's pymeshlab quadric edge-collapse
 * decimation to at most 40 000 faces (preserve boundary / normals / topology).  HOST pointers, synchronous, no GPU
 * involved (the reference's is MeshLab's serial CPU code; it runs once per image).  verts (V,3) float32, faces (F,3)
 * int64 -> out_verts (room for V x 3), out_faces (room for F x 3), out_counts[2] = {vertices, faces} written.
 * Vertices and faces keep their relative order; the same input gives the same output. */
FOHO_API int foho_mesh_decimate(const float* verts, int32_t V, const int64_t* faces, int32_t F, int32_t target_faces,
                       float* out_verts, int64_t* out_faces, int32_t* out_counts);

/* ---- the ShapeVAE geometry decoder of latent2sdf on the matrix cores ----------------------
 * Replaces the loop `for start in range(0, N, 8000): logits.append(vae.geo_decoder(queries, latents))` of latent2sdf
 * (PL:298-308; hy3dgen CrossAttentionDecoder: FourierEmbedder -> query_proj -> ResidualCrossAttentionBlock over the
 * latent tokens -> ln_post -> output_proj), which the reference runs 550 times per image (PL:1391-1393, 1507-1509) plus
 * once per denoising step without gradients (PL:1624-1642, 385^3 queries on the last one).  Forward only in this version.
 * fp16 storage / fp32 accumulation (the reference runs the VAE in fp16, PL:522); all queries in ONE call.
 * Weights: DEVICE pointers, prepared once by the caller.  fp16 matrices in torch.nn.Linear layout (out_features rows of
 * in_features), fp32 biases / LayerNorm parameters.  Constraints (FOHO_ERR_BAD_ARG otherwise): head dimension 64
 * (width = 64 heads), width % 128 == 0 and <= 1024, n_latents % 64 == 0, hidden % 128 == 0, n_freqs <= 10. */
typedef struct {
    int32_t width, heads, n_latents, hidden, n_freqs;
    int32_t flags;                   /* FOHO_GEO_* bits below; 0 = the product path */
    const float* freqs;              /* (n_freqs) DEVICE: the embedder's frequencies (2^j pi, or 2^j with include_pi = False) */
    const void* w_qproj;             /* (width, 64) fp16: query_proj, columns 3 (2 n_freqs + 1) .. 63 zero      */
    const float* b_qproj;            /* (width)                                                                 */
    const float *ln_q_g, *ln_q_b;    /* LayerNorm of the queries (ln_1), (width) each                           */
    const float *ln_kv_g, *ln_kv_b;  /* LayerNorm of the latent tokens (ln_2 of the cross-attention block)      */
    const void* w_q;                 /* (width, width) fp16: c_q                                                */
    const float* b_q;                /* (width)  (zeros when the model has qkv_bias = False)                    */
    const void* w_kv;                /* (2 width, width) fp16: c_kv with rows ordered [K of head 0 .. K of head h-1 | V of head 0 ..]
                                        (hy3dgen interleaves K and V per head; the caller permutes the rows once)  */
    const float* b_kv;               /* (2 width)                                                               */
    const void* w_proj;              /* (width, width) fp16: c_proj                                             */
    const float* b_proj;
    const float *ln_2_g, *ln_2_b;    /* LayerNorm before the MLP (ln_3 in hy3dgen)                              */
    const void* w_fc1;               /* (hidden, width) fp16; GELU (erf form) behind it                         */
    const float* b_fc1;
    const void* w_fc2;               /* (width, hidden) fp16                                                    */
    const float* b_fc2;
    const float *ln_post_g, *ln_post_b;
    const float* w_out;              /* (width) fp32: output_proj (one occupancy logit)                         */
    float b_out;
    float ln_eps;                    /* eps of ln_post (and of the other LayerNorms while their own fields below are 0) */
    float prior_radius, prior_sharpness, out_gain;  /* logits = sharpness (radius - |x|) + gain * learned.  A trained decoder:
                                        0, 0, 1.  (The random-initialised stand-in of the tests adds an analytic sphere.) */
    /* backward only (foho_geo_decode_bwd; may be NULL for forward-only use): the three matrices the gradient flows back
     * through, TRANSPOSED once by the caller, and a buffer of zeros (the backward GEMMs have no bias)               */
    const void* w_fc2_t;             /* (hidden, width) fp16 = w_fc2^T                                          */
    const void* w_fc1_t;             /* (width, hidden) fp16 = w_fc1^T                                          */
    const void* w_proj_t;            /* (width, width) fp16 = w_proj^T                                          */
    const float* zeros;              /* (max(hidden, width)) fp32 zeros                                         */
    /* hy3dgen's qk_norm (attention_blocks.py: LayerNorm over the head dimension on q and on k, before the scaled dot
     * product): 129 floats each -- gain (64), bias (64), eps -- or NULL for a decoder without it.  foho_geo_prepare
     * normalises K; K / V handed to foho_geo_set_kv are expected normalised already (the caller's autograd owns that).     */
    const float* q_norm;
    const float* k_norm;
    /* version 104: eps of the three LayerNorms in front of the attention / the MLP (hy3dgen builds the block's ln_1 / ln_2 / ln_3
     * with eps 1e-6 and ln_post with torch's default 1e-5); 0 = the same as ln_eps, which stays ln_post's                      */
    float ln_q_eps, ln_kv_eps, ln_2_eps, reserved2;
} foho_geo_weights;
/* foho_geo_weights.flags.  FOHO_GEO_NO_LNFUSE: run the forward chain with its two LayerNorm kernels instead of the folded form (ln_2
 * inside fc1, ln_post + output_proj inside fc2's epilogue): same logits to one fp16 ulp, 3 % slower -- for A/B measurements and the
 * parity test of the folded form.  (Until version 104 this was the environment variable FOHO_GEO_LNFUSE, read at every call.) */
#define FOHO_GEO_NO_LNFUSE 1

/* sizeof(foho_geo_weights) of the loaded build (version 104): the Python side compares it with its ctypes mirror before the
 * first call, like foho_abi_sizes does for the step's structs */
FOHO_API int64_t foho_geo_abi_size(void);
/* workspace for row blocks of `chunk_rows` queries: K / V of the latent tokens + the block's activations (14.5 KB per
 * query at width 1024 / hidden 4096; 16384 rows keep a block inside the 256 MB Infinity Cache).  0 on bad arguments. */
FOHO_API size_t foho_geo_workspace_bytes(const foho_geo_weights* w, int32_t chunk_rows);
/* once per set of latent tokens: LayerNorm + K/V projection of `latents` (n_latents, width) fp16 into the workspace
 * (V transposed and key-permuted for the attention kernel).  The same chunk_rows as the decode calls that follow.
 * Also rebuilds, from the weights as they are at this call, the operands of the forward's folded LayerNorms (fc1's weights
 * times ln_2's gain, their row sums, ln_post's gain times w_out: 8 MB, two small kernels) -- foho_geo_decode_fwd[_cached] run
 * ln_2 inside fc1 and ln_post + output_proj inside fc2's epilogue; weights changed AFTER the prepare call take effect with
 * the next one.  flags & FOHO_GEO_NO_LNFUSE runs the chain with its LayerNorm kernels instead (same logits to one fp16 ulp). */
FOHO_API int foho_geo_prepare(const foho_geo_weights* w, const void* latents, int32_t chunk_rows, void* workspace, size_t workspace_bytes,
                     void* stream);
/* logits[n] = geo_decoder(queries[n], latents) for n < n_queries: queries (N,3) fp32 (already rounded the way the caller's
 * pipeline rounds them: the reference casts them to fp16 first, PL:303), logits (N) fp32.  Uses the K / V that
 * foho_geo_prepare left in the workspace.  9 launches per row block, asynchronous, no host synchronisation. */
FOHO_API int foho_geo_decode_fwd(const foho_geo_weights* w, const float* queries, int64_t n_queries, float* logits, int32_t chunk_rows,
                        void* workspace, size_t workspace_bytes, void* stream);
/* Gradients to the latent tokens: the decoder as a differentiable function of K / V.
 * foho_geo_set_kv installs K / V computed by the caller -- kv (n_latents, 2 width) fp16 = c_kv(ln(latents)), rows [K of all
 * heads | V of all heads]: the caller's autograd owns LayerNorm + projection of the 3072 tokens (0.1 % of the work) -- into the
 * forward workspace, in place of foho_geo_prepare.  foho_geo_decode_bwd then returns grad_kv (n_latents, 2 width) fp32 =
 * d sum(grad_logits . logits) / d kv for the logits the forward computes from those K / V and `queries`, running the chain
 * backwards per row block -- LayerNorm / GELU backward, three GEMMs with the transposed weights, the attention backward for K and
 * V (queries are constants: nothing flows to them); no atomics: partial sums per (split of the row tiles, key block), added in a
 * fixed order, so the result is bitwise repeatable.  What the backward needs from the forward (pre-activation, attention output and
 * log-sum-exp, the scaled queries in both orientations, the two residual streams: 18.1 KB per query at width 1024) comes
 *   - from `saved`, filled by foho_geo_decode_fwd_keep (the forward to call when a backward will follow; foho_geo_saved_bytes:
 *     5 GB for a 65^3 grid -- the part has 288), or
 *   - with saved = NULL, from a RECOMPUTATION of the forward chain per row block (+11 ms per 65^3 grid, no memory).
 * `bwd_workspace`: foho_geo_bwd_workspace_bytes(w, chunk_rows), used by both calls. */
FOHO_API size_t foho_geo_bwd_workspace_bytes(const foho_geo_weights* w, int32_t chunk_rows);
FOHO_API size_t foho_geo_saved_bytes(const foho_geo_weights* w, int32_t chunk_rows, int64_t n_queries);
FOHO_API int foho_geo_set_kv(const foho_geo_weights* w, const void* kv, int32_t chunk_rows, void* workspace, size_t workspace_bytes, void* stream);
FOHO_API int foho_geo_decode_fwd_keep(const foho_geo_weights* w, const float* queries, int64_t n_queries, float* logits, int32_t chunk_rows,
                             void* workspace, size_t workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes, void* saved,
                             size_t saved_bytes, void* stream);
FOHO_API int foho_geo_decode_bwd(const foho_geo_weights* w, const float* queries, int64_t n_queries, const float* grad_logits, float* grad_kv,
                        int32_t chunk_rows, void* workspace, size_t workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes,
                        const void* saved, size_t saved_bytes, void* stream);
/* The query side, cached.  The grid the latent is decoded on never changes during a guidance run (PL:1125-1143: 65^3 points for
 * all 550 inner iterations of every image) and a quarter of the forward chain depends on nothing else: Fourier embedding ->
 * query_proj -> ln_1 -> c_q (-> q_norm) (PL:298-308 runs it again per chunk and per decode).  foho_geo_prepare_queries computes
 * x0 = query_proj(embed(q)) and the scaled attention queries of all rows once into `cache` (foho_geo_query_cache_bytes: 4 KB per
 * query at width 1024 -- 1.1 GB per 65^3 grid); foho_geo_decode_fwd_cached is foho_geo_decode_fwd from there on: the same kernels
 * on the same numbers, logits bitwise equal. */
FOHO_API size_t foho_geo_query_cache_bytes(const foho_geo_weights* w, int64_t n_queries);
FOHO_API int foho_geo_prepare_queries(const foho_geo_weights* w, const float* queries, int64_t n_queries, int32_t chunk_rows, void* workspace,
                             size_t workspace_bytes, void* cache, size_t cache_bytes, void* stream);
FOHO_API int foho_geo_decode_fwd_cached(const foho_geo_weights* w, const float* queries, int64_t n_queries, const void* cache, size_t cache_bytes,
                               float* logits, int32_t chunk_rows, void* workspace, size_t workspace_bytes, void* stream);
/* foho_geo_decode_bwd over the ACTIVE rows only.  The gradient that reaches latent2sdf in the guidance loop comes out of the
 * FlexiCubes backward (PL:1507-1509, 1600) and is non-zero only at the end points of the grid edges the iso-surface crosses
 * (5-10 % of a 65^3 grid); a row with a zero logit gradient adds exactly zero to dK / dV.  The rows with grad_logits != 0 are
 * compacted on the device (in row order: the result is bitwise repeatable), the forward chain is recomputed for them and the
 * backward runs on them -- the forward in front of it is plain foho_geo_decode_fwd[_cached]: nothing is kept.  The number of
 * active rows stays in device memory: launches are sized for row blocks of the capacity, each kernel works on
 * min(block, what is left of the count) rows and row blocks beyond the count leave at once: no host synchronisation, capturable in
 * a hipGraph.  row_cap: the caller's upper bound on the number of active rows (<= 0 or > n_queries: n_queries, which can never
 * overflow); rows beyond it are dropped and COUNTED.  stats_out: optional DEVICE int32[2] = {active rows, rows dropped}.
 * rows_workspace: foho_geo_rows_workspace_bytes(n_queries, row_cap, chunk_rows); workspace / bwd_workspace as for foho_geo_decode_bwd. */
FOHO_API size_t foho_geo_rows_workspace_bytes(int64_t n_queries, int64_t row_cap, int32_t chunk_rows);
FOHO_API int foho_geo_decode_bwd_rows(const foho_geo_weights* w, const float* queries, int64_t n_queries, const float* grad_logits, float* grad_kv,
                             int64_t row_cap, int32_t chunk_rows, void* workspace, size_t workspace_bytes, void* bwd_workspace,
                             size_t bwd_workspace_bytes, void* rows_workspace, size_t rows_workspace_bytes, int32_t* stats_out, void* stream);
/* Attention as an operator, forward and backward (version 104): O = softmax(Q K^T / sqrt(64)) V per head of 64 -- what
 * torch.nn.functional.scaled_dot_product_attention computes (no mask, no dropout) for the self-attention layers of the ShapeVAE
 * transformer that latent2sdf runs and back-propagates in front of the geometry decoder in every inner iteration (PL:295, 1391-1393,
 * 1507-1509).  The kernels are the decoder's own (k_geo_attn; k_geo_attn_bwd for dK / dV) plus k_geo_attn_dq.
 * q, k, v: fp16, read WHERE THEY LIE -- element (batch b, head h, row n, d) of q at q + b q_batch + n q_row + h q_head + d (k and v
 * share kv_batch / kv_row / kv_head); that covers (B, N, H, 64) tensors viewed as (B, H, N, 64), the [K | V] output of one Linear
 * (kv_row = 2 x 64 H, v = k + 64 H) and hy3dgen's interleaved q | k | v per head (row 3 x 64 H, head 192).  The 1 / sqrt(64) scale
 * and the log2(e) of the exp2 softmax are applied inside.  out (batch, M, 64 H) fp16, heads side by side; nlse (batch, M rounded up
 * to 64, H) fp32, written by the forward and handed back to the backward (either may be NULL); lse_natural (batch, H, M) fp32: the
 * natural-log log-sum-exp of the scaled scores, the form torch's own attention backward takes; grad_out / grad_q like out, grad_k / grad_v (batch, L,
 * 64 H) fp16.  L a multiple of 64 (128 for the backward), 1..16 heads; one workspace (foho_sdpa_workspace_bytes) serves any batch.
 * followmyhold_amd.sdpa wraps them as an autograd function and can stand in for F.scaled_dot_product_attention inside a context. */
typedef struct foho_sdpa_desc {
    int32_t M, L, heads, batch;           /* queries, keys, heads of 64, batch items */
    int64_t q_batch, q_row, q_head;       /* strides of q in halfs */
    int64_t kv_batch, kv_row, kv_head;    /* strides of k and of v in halfs */
} foho_sdpa_desc;
FOHO_API size_t foho_sdpa_workspace_bytes(int32_t M, int32_t L, int32_t heads);
FOHO_API int foho_sdpa_fwd(const foho_sdpa_desc* d, const void* q, const void* k, const void* v, void* out, float* nlse, float* lse_natural,
                  void* workspace, size_t workspace_bytes, void* stream);
FOHO_API int foho_sdpa_bwd(const foho_sdpa_desc* d, const void* q, const void* k, const void* v, const void* out, const float* nlse,
                  const void* grad_out, void* grad_q, void* grad_k, void* grad_v, void* workspace, size_t workspace_bytes, void* stream);
/* ---- the ShapeVAE transformer of latent2sdf, forward and backward to its input (version 105) ----------------------------
 * Replaces `pred = vae(pred)` of latent2sdf (PL:295; hy3dgen ShapeVAE.forward = post_kl -> Transformer: n_layers
 * ResidualAttentionBlocks  x = x + c_proj(attention(qk_norm(c_qkv(ln_1(x)))));  x = x + mlp.c_proj(gelu(mlp.c_fc(ln_2(x))))  over the
 * 3072 latent tokens), which the reference runs AND back-propagates in every one of the 550 inner iterations per image
 * (PL:1391-1393, 1507-1509, 1600: the guidance gradient reaches the noise prediction through it).  The weights are constants of the
 * guidance (only the noise prediction and the pose are optimised): the backward returns the gradient with respect to the INPUT tokens
 * only.  post_kl (a 64 -> width Linear) stays with the caller.
 * x0 / out / grad_out / grad_x0: (batch x n_tokens, width) fp16, images one after the other.  fp16 storage, fp32 accumulation.
 * Per layer 4 GEMMs + the attention kernels: both LayerNorms are FOLDED into the GEMM behind them (the GEMM runs on the
 * un-normalised rows with gamma folded into the weights; mean / rstd per row -- left behind by the epilogue of the GEMM in front --
 * are applied in the epilogue), GELU, residuals, qk_norm and the statistics are epilogues.  The caller prepares, once:
 *   w_qkv     (3 width, width) fp16 = rows [Q of all heads | K of all heads | V of all heads] of c_qkv, each times ln_1's gain
 *             (hy3dgen interleaves q | k | v per head: the caller permutes the rows)
 *   fold_qkv  fp32: [b' (3 width) = bias + W beta | s (3 width) = row sums of the ROUNDED folded weights |
 *             q_norm: gain (64), bias (64), eps, 3 pad | k_norm: the same]  (the last 264 only when qk_norm != 0)
 *   w_qkv_t   (width, 3 width) fp16 = w_qkv^T (of the folded weights: the backward GEMM then yields d / d xhat directly)
 *   w_proj (width, width), b_proj, w_proj_t = w_proj^T
 *   w_fc1 (hidden, width) folded with ln_2's gain, fold_fc1 = [b' (hidden) | s (hidden)], w_fc1_t = w_fc1^T (folded)
 *   w_fc2 (width, hidden), b_fc2, w_fc2_t = w_fc2^T
 * Constraints: head dimension 64, width % 128 == 0 and <= 1024, hidden % 128 == 0, n_tokens % 128 == 0. */
typedef struct foho_vae_layer {
    const void* w_qkv; const float* fold_qkv; const void* w_qkv_t;
    const void* w_proj; const float* b_proj; const void* w_proj_t;
    const void* w_fc1; const float* fold_fc1; const void* w_fc1_t;
    const void* w_fc2; const float* b_fc2; const void* w_fc2_t;
    float eps1, eps2;                 /* of ln_1 / ln_2 */
    int32_t qk_norm, reserved;        /* != 0: LayerNorm over the 64 head dimensions of q and of k (parameters at the end of fold_qkv) */
} foho_vae_layer;
typedef struct foho_vae_desc {
    int32_t width, heads, hidden, n_layers, n_tokens, batch;
    const foho_vae_layer* layers;     /* HOST array of n_layers records (the pointers inside are DEVICE pointers) */
    const float* zeros;               /* DEVICE: max(hidden, 3 width) fp32 zeros (the backward GEMMs have no bias) */
    int32_t flags, reserved;          /* FOHO_VAE_* bits below; 0 = the product path */
} foho_vae_desc;
/* foho_vae_desc.flags (A/B measurements; results agree to the rounding of the sums).  FOHO_VAE_SPLIT_DKV: the attention backward's dK / dV
 * as partial sums over three splits of the query tiles + a reduction pass (the form foho_sdpa_bwd uses) instead of one workgroup per
 * (head, key block) writing its sums directly. */
#define FOHO_VAE_SPLIT_DKV 1
/* sizeof(foho_vae_layer) * 1000 + sizeof(foho_vae_desc): bindings that mirror the structs compare it with their own */
FOHO_API int64_t foho_vae_abi_size(void);
/* scratch of either direction / what foho_vae_fwd keeps for foho_vae_bwd (per layer: input, q | k | v (+ their un-normalised copy
 * with qk_norm), attention output and log-sum-exp, the MLP's input and pre-activation: 65-84 MB at 3072 tokens x 1024).  0 on bad arguments. */
FOHO_API size_t foho_vae_workspace_bytes(const foho_vae_desc* d);
FOHO_API size_t foho_vae_saved_bytes(const foho_vae_desc* d);
/* out = transformer(x0).  saved == NULL: nothing is kept (inference).  Asynchronous, 8 launches per layer and image, no host synchronisation. */
FOHO_API int foho_vae_fwd(const foho_vae_desc* d, const void* x0, void* out, void* workspace, size_t workspace_bytes, void* saved, size_t saved_bytes,
                          void* stream);
/* grad_x0 = d sum(grad_out . out) / d x0 for the forward that filled `saved`. */
FOHO_API int foho_vae_bwd(const foho_vae_desc* d, const void* grad_out, void* grad_x0, void* workspace, size_t workspace_bytes, const void* saved,
                          size_t saved_bytes, void* stream);
/* building blocks on their own (unit tests, profiling).  foho_geo_gemm: C (M,N) fp16 = epilogue(A (M,K) . Wt (N,K)^T + bias)
 * with epilogue = GELU when `gelu & 1`, x scale, + R (M,N) when R is not NULL (not both); N % 128 == 0, K % 64 == 0.
 * Shapes with N % 256 == 0, K >= 256 and M >= 2048 run on 256 x 256 tiles (the phased, persistent kernel) unless `gelu & 2` asks
 * for the 128 x 128 kernel, `gelu & 4` for the lock-step 256 x 256 one, `gelu & 8` for the 128 x 128 kernel with the four-deep ring, `gelu & 32`
 * for its eight-wave form with fill waves and matrix waves (what launches of at most one workgroup per CU get), `gelu & 16` for the phased one,
 * `gelu & 64` for the phased one on 192 x 256 tiles (what a single, under-filled round of 256-row tiles gets) (arguments of THIS call: the library keeps no mode state).
 * foho_geo_attention: O (M, 64 heads) = softmax(Q K^T) V per head with Q (M, 64 heads) pre-scaled by log2(e) / 8, KV
 * (n_latents, 128 heads) = [K | V] as the projection leaves them, Vt_scratch room for 64 heads x n_latents fp16. */
FOHO_API int foho_geo_gemm(const void* A, const void* Wt, const float* bias, const void* R, void* C, int32_t M, int32_t N, int32_t K,
                  int32_t gelu, float scale, void* stream);
FOHO_API int foho_geo_attention(const void* Q, const void* KV, void* Vt_scratch, void* O, int32_t M, int32_t n_latents, int32_t heads,
                       void* stream);
FOHO_API const char* foho_geo_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* FOHO_HIP_H */
