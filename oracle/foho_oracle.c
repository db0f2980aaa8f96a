/*
 * foho_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the discrete / per-element geometry operators that
 * FollowMyHold's guidance hot path delegates to un-vendored pytorch3d / kaolin:
 *
 *   - naive mesh rasteriser (pytorch3d RasterizeMeshesNaive semantics, used by
 *     MeshRasterizer(K=1, bin_size=-1) at reference src/foho/guidance/run.py:95-105
 *     and K=100 at run.py:106-116; called from
 *     third_party_patches/hy3dgen/shapegen/pipelines.py:272-274,1328-1329,1488,1546-1549)
 *   - watertight-mesh inside test (kaolin check_sign semantics: ray parity;
 *     reference third_party/utilz/kaolin_sdf_ops.py:104)
 *   - exact point->triangle squared distance (kaolin point_to_mesh_distance;
 *     kaolin_sdf_ops.py:101)
 *   - brute-force K=1 nearest neighbour (pytorch3d knn_points; pipelines.py:1529-1538)
 *
 * PARITY UNPINNED: pytorch3d (git HEAD, unpinned) and kaolin==0.17.0 are not
 * vendored under /root/reference and cannot be installed here, so this file
 * restates their published algorithms and is pinned only
 * by analytic known-answer tests (tests/test_oracle_kat.py), not by reference
 * golden vectors.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Build: `make -C oracle` (gcc -O2 -ffp-contract=off: every
 * float op is a separately rounded IEEE-754 binary32 op, the same contract the
 * HIP kernels are compiled under, so face indices can be compared bit-exactly).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define K_EPS 1e-8f

/* ------------------------------------------------------------------------- */
/* rasteriser primitives                             */
/* ------------------------------------------------------------------------- */

static inline float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

/* pixel index -> NDC coordinate of the pixel centre; S1 is the size of the
 * axis being converted, S2 the other one (non-square images span a wider NDC
 * range along the longer axis). */
static inline float pix_to_ndc(int i, int S1, int S2) {
    float range = 2.0f;
    if (S1 > S2) range = ((float)S1 * range) / (float)S2;
    const float offset = range / 2.0f;
    return -offset + (range * (float)i + offset) / (float)S1;
}

static inline float seg_d2(float px, float py, float ax, float ay, float bx, float by) {
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

typedef struct {
    float z;
    int64_t face;   /* index into the caller's (unclipped) face list */
    float sdist;
    float b0, b1, b2;
    int32_t sub;    /* -1: the face itself; 0 / 1: which sub-triangle of a face clipped at the near plane */
    int64_t cidx;   /* index into the clipped face list (neighbour matching) */
} frag_t;

/* Evaluate one (pixel centre, face) pair.  Returns 1 when the face produces a
 * fragment at this pixel. */
static inline int eval_pixel_face(const float* fv, float xf, float yf, float blur_radius,
                                  float sqrt_blur, int perspective_correct, int clip_bary,
                                  int cull_backfaces, frag_t* out) {
    const float x0 = fv[0], y0 = fv[1], z0 = fv[2];
    const float x1 = fv[3], y1 = fv[4], z1 = fv[5];
    const float x2 = fv[6], y2 = fv[7], z2 = fv[8];

    const float zmax = fmaxf(fmaxf(z0, z1), z2);
    if (zmax < 0.0f) return 0;
    const float xmin = fminf(fminf(x0, x1), x2) - sqrt_blur;
    const float xmax = fmaxf(fmaxf(x0, x1), x2) + sqrt_blur;
    const float ymin = fminf(fminf(y0, y1), y2) - sqrt_blur;
    const float ymax = fmaxf(fmaxf(y0, y1), y2) + sqrt_blur;
    if (!(xmin <= xf && xf <= xmax && ymin <= yf && yf <= ymax)) return 0;
    const float face_area = edge_fn(x0, y0, x1, y1, x2, y2);
    if (cull_backfaces && face_area < 0.0f) return 0;
    if (face_area <= K_EPS && face_area >= -K_EPS) return 0;

    const float area = edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS;
    const float a0 = edge_fn(xf, yf, x1, y1, x2, y2) / area;
    const float a1 = edge_fn(xf, yf, x2, y2, x0, y0) / area;
    const float a2 = edge_fn(xf, yf, x0, y0, x1, y1) / area;

    float w0 = a0, w1 = a1, w2 = a2;
    if (perspective_correct) {
        const float t0 = a0 * z1 * z2;
        const float t1 = z0 * a1 * z2;
        const float t2 = z0 * z1 * a2;
        const float den = fmaxf(t0 + t1 + t2, K_EPS);
        w0 = t0 / den;
        w1 = t1 / den;
        w2 = t2 / den;
    }
    float c0 = w0, c1 = w1, c2 = w2;
    if (clip_bary) {
        c0 = fmaxf(w0, 0.0f);
        c1 = fmaxf(w1, 0.0f);
        c2 = fmaxf(w2, 0.0f);
        const float s = fmaxf(c0 + c1 + c2, 1e-5f);
        c0 = c0 / s;
        c1 = c1 / s;
        c2 = c2 / s;
    }
    const float pz = c0 * z0 + c1 * z1 + c2 * z2;
    if (pz < 0.0f) return 0;

    const float d01 = seg_d2(xf, yf, x0, y0, x1, y1);
    const float d02 = seg_d2(xf, yf, x0, y0, x2, y2);
    const float d12 = seg_d2(xf, yf, x1, y1, x2, y2);
    const float dist = fminf(fminf(d01, d02), d12);
    const int inside = (w0 > 0.0f) && (w1 > 0.0f) && (w2 > 0.0f);
    if (!inside && dist >= blur_radius) return 0;

    out->z = pz;
    out->sdist = inside ? -dist : dist;
    out->b0 = c0;
    out->b1 = c1;
    out->b2 = c2;
    return 1;
}

/* K-buffer insertion identical to the naive rasteriser: fill, then replace the
 * current farthest entry when strictly nearer. */
static inline void kbuf_insert(frag_t* q, int* q_size, float* q_max_z, int* q_max_idx, int K,
                               const frag_t* f) {
    if (*q_size < K) {
        q[*q_size] = *f;
        if (f->z > *q_max_z) {
            *q_max_z = f->z;
            *q_max_idx = *q_size;
        }
        (*q_size)++;
    } else if (f->z < *q_max_z) {
        q[*q_max_idx] = *f;
        *q_max_z = f->z;
        for (int i = 0; i < K; i++) {
            if (q[i].z > *q_max_z) {
                *q_max_z = q[i].z;
                *q_max_idx = i;
            }
        }
    }
}

static inline void kbuf_sort(frag_t* q, int n) { /* stable bubble sort by z */
    for (int i = 0; i < n - 1; i++)
        for (int j = 0; j < n - 1 - i; j++)
            if (q[j + 1].z < q[j].z) {
                frag_t t = q[j];
                q[j] = q[j + 1];
                q[j + 1] = t;
            }
}

/* Near plane.  MeshRasterizer passes z_clip_value = znear / 2 to rasterize_meshes for perspective cameras
 * (raster_settings.z_clip_value is None at reference src/foho/guidance/run.py:95-105; znear = 0.01, run.py:84-90) and
 * rasterize_meshes runs pytorch3d.renderer.mesh.clip.clip_faces on the (x_ndc, y_ndc, z_view) face vertices first.
 * Restated here from the published algorithm (pytorch3d is not vendored: parity unpinned), per face, with
 * clipped_i = (z_i < z_clip):
 *   0 clipped   the face is rasterised as it is
 *   3 clipped   culled
 *   2 clipped   p1 = the vertex in front, p2 / p3 = the next two in cyclic order; p4 / p5 = where the edges p1p2 / p1p3
 *               cross the plane; ONE triangle (p4, p5, p1)
 *   1 clipped   p1 = the vertex behind, p2 / p3 as above; the remaining quadrilateral p4 p2 p3 p5 is split into TWO
 *               triangles (p4, p2, p5) and (p5, p2, p3), consecutive in the clipped list and each other's "neighbour"
 *   crossing    w = (z1 - c) / (z1 - z_other); z = z1 (1 - w) + z_other w; perspective cameras interpolate x z and y z
 *               ("world" xy) and divide by c, so the cut is straight in space, not in the image
 * The rasteriser then runs over the clipped list.  A pixel can receive a fragment from at most one of two neighbours: when
 * the second one arrives while the first sits in the pixel's K-buffer, it REPLACES it if its unsigned edge distance is
 * smaller and is dropped otherwise (rasterize_meshes' CheckPixelInsideFace).  Face ids are mapped back to the caller's
 * faces and barycentrics to the unclipped face (bary_unclipped = bary_sub . barycentrics of the sub-triangle's vertices).
 * The default plane is the path's only camera's; tests may move or disable (z_clip < -1e30) it. */
static float g_z_clip = 0.01f * 0.5f;
void foho_oracle_set_z_clip(float z) { g_z_clip = z; }
float foho_oracle_get_z_clip(void) { return g_z_clip; }
int64_t foho_oracle_count_near_clipped(const float* face_verts, int64_t F) {
    int64_t n = 0;
    for (int64_t f = 0; f < F; f++) {
        const float* v = face_verts + 9 * f;
        const float zmin = fminf(fminf(v[2], v[5]), v[8]), zmax = fmaxf(fmaxf(v[2], v[5]), v[8]);
        if (zmin < g_z_clip && !(zmax < g_z_clip)) n++;
    }
    return n;
}

typedef struct {
    float v[9];       /* the triangle that is rasterised */
    float bc[9];      /* row k: barycentrics of its vertex k w.r.t. the unclipped face */
    int64_t face;     /* unclipped face */
    int64_t neighbor; /* index of the other half of a split face in the clipped list, -1 otherwise */
    int32_t sub;
} cface_t;

/* point where the edge p1 -> po crosses z = c, and its barycentrics (i1, io = positions of p1, po in the face) */
static inline void plane_crossing(const float* p1, const float* po, int i1, int io, float c, int perspective, float* out,
                                  float* bc) {
    const float w = (p1[2] - c) / (p1[2] - po[2]);
    const float u = 1.0f - w;
    out[0] = p1[0] * u + po[0] * w;
    out[1] = p1[1] * u + po[1] * w;
    out[2] = p1[2] * u + po[2] * w;
    if (perspective) {
        out[0] = ((p1[0] * p1[2]) * u + (po[0] * po[2]) * w) / c;
        out[1] = ((p1[1] * p1[2]) * u + (po[1] * po[2]) * w) / c;
    }
    bc[0] = bc[1] = bc[2] = 0.0f;
    bc[i1] = u;
    bc[io] = w;
}

/* sub-triangles of one face: returns their number (0: culled, 1 or 2), -1 when the face is not clipped at all */
static inline int clip_one_face(const float* fv, float c, int perspective, float tri[2][9], float bc[2][9]) {
    const int b0 = fv[2] < c, b1 = fv[5] < c, b2 = fv[8] < c;
    const int n = b0 + b1 + b2;
    if (n == 0) return -1;
    if (n == 3) return 0;
    /* p1: the one vertex on its side of the plane */
    const int i1 = (n == 2) ? (!b0 ? 0 : (!b1 ? 1 : 2)) : (b0 ? 0 : (b1 ? 1 : 2));
    const int i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
    const float *p1 = fv + 3 * i1, *p2 = fv + 3 * i2, *p3 = fv + 3 * i3;
    float p4[3], p5[3], b4[3], b5[3], e1[3] = {0, 0, 0}, e2[3] = {0, 0, 0}, e3[3] = {0, 0, 0};
    plane_crossing(p1, p2, i1, i2, c, perspective, p4, b4);
    plane_crossing(p1, p3, i1, i3, c, perspective, p5, b5);
    e1[i1] = e2[i2] = e3[i3] = 1.0f;
#define PUT(t, k, P, B)                                  \
    do {                                                 \
        for (int q_ = 0; q_ < 3; q_++) {                 \
            tri[t][3 * (k) + q_] = (P)[q_];              \
            bc[t][3 * (k) + q_] = (B)[q_];               \
        }                                                \
    } while (0)
    if (n == 2) {
        PUT(0, 0, p4, b4);
        PUT(0, 1, p5, b5);
        PUT(0, 2, p1, e1);
        return 1;
    }
    PUT(0, 0, p4, b4);
    PUT(0, 1, p2, e2);
    PUT(0, 2, p5, b5);
    PUT(1, 0, p5, b5);
    PUT(1, 1, p2, e2);
    PUT(1, 2, p3, e3);
#undef PUT
    return 2;
}

static cface_t* clip_faces(const float* face_verts, int64_t F, int perspective, int64_t* n_out) {
    cface_t* out = (cface_t*)malloc(sizeof(cface_t) * (size_t)(2 * F + 1));
    int64_t n = 0;
    const int enabled = g_z_clip > -1e30f;
    for (int64_t f = 0; f < F; f++) {
        float tri[2][9], bc[2][9];
        const int k = enabled ? clip_one_face(face_verts + 9 * f, g_z_clip, perspective, tri, bc) : -1;
        if (k == 0) continue;
        if (k < 0) {
            cface_t* c = &out[n++];
            memcpy(c->v, face_verts + 9 * f, sizeof(float) * 9);
            for (int q = 0; q < 9; q++) c->bc[q] = (q % 4 == 0) ? 1.0f : 0.0f;
            c->face = f;
            c->neighbor = -1;
            c->sub = -1;
            continue;
        }
        for (int t = 0; t < k; t++) {
            cface_t* c = &out[n + t];
            memcpy(c->v, tri[t], sizeof(float) * 9);
            memcpy(c->bc, bc[t], sizeof(float) * 9);
            c->face = f;
            c->sub = t;
            c->neighbor = (k == 2) ? n + (1 - t) : -1;
        }
        n += k;
    }
    *n_out = n;
    return out;
}

/* Per-entry screen boxes (inflated by sqrt(blur)); used only to skip eval_pixel_face() early -- it re-tests exactly the
 * same box. */
static float* make_face_boxes(const cface_t* cf, int64_t n, float sqrt_blur) {
    float* box = (float*)malloc(sizeof(float) * 4 * (n > 0 ? n : 1));
    for (int64_t f = 0; f < n; f++) {
        const float* v = cf[f].v;
        box[4 * f + 0] = fminf(fminf(v[0], v[3]), v[6]) - sqrt_blur;
        box[4 * f + 1] = fmaxf(fmaxf(v[0], v[3]), v[6]) + sqrt_blur;
        box[4 * f + 2] = fminf(fminf(v[1], v[4]), v[7]) - sqrt_blur;
        box[4 * f + 3] = fmaxf(fmaxf(v[1], v[4]), v[7]) + sqrt_blur;
    }
    return box;
}

/* one fragment of clipped entry ci into the pixel's K-buffer (CheckPixelInsideFace's neighbour rule, then kbuf_insert);
 * barycentrics are converted to the unclipped face on the way in */
static inline void kbuf_insert_entry(frag_t* q, int* q_size, float* q_max_z, int* q_max_idx, int K, frag_t* fr,
                                     const cface_t* cf, int64_t ci) {
    const cface_t* c = &cf[ci];
    fr->face = c->face;
    fr->sub = c->sub;
    fr->cidx = ci;
    if (c->sub >= 0) {
        const float s0 = fr->b0, s1 = fr->b1, s2 = fr->b2;
        fr->b0 = (s0 * c->bc[0] + s1 * c->bc[3]) + s2 * c->bc[6];
        fr->b1 = (s0 * c->bc[1] + s1 * c->bc[4]) + s2 * c->bc[7];
        fr->b2 = (s0 * c->bc[2] + s1 * c->bc[5]) + s2 * c->bc[8];
    }
    if (c->neighbor >= 0) {
        for (int i = 0; i < *q_size; i++)
            if (q[i].cidx == c->neighbor) {
                if (fabsf(fr->sdist) < fabsf(q[i].sdist)) {
                    q[i] = *fr;
                    if (fr->z > *q_max_z) {
                        *q_max_z = fr->z;
                        *q_max_idx = i;
                    }
                }
                return;
            }
    }
    kbuf_insert(q, q_size, q_max_z, q_max_idx, K, fr);
}

/*
 * Naive rasteriser with materialised K-buffer.
 * face_verts: (F,3,3) rows (x_ndc, y_ndc, z_view).  Outputs are (H,W,K[,3]),
 * background pix_to_face=-1, zbuf=-1, bary=-1, dists=-1.
 */
int foho_oracle_rasterize(const float* face_verts, int64_t F, int H, int W, float blur_radius,
                          int K, int perspective_correct, int clip_bary, int cull_backfaces,
                          int64_t* pix_to_face, float* zbuf, float* bary, float* dists) {
    if (K < 1 || K > 256) return -1;
    const float sqrt_blur = sqrtf(blur_radius);
    int64_t nc = 0;
    cface_t* cf = clip_faces(face_verts, F, perspective_correct, &nc);
    float* box = make_face_boxes(cf, nc, sqrt_blur);
#pragma omp parallel for schedule(dynamic, 4)
    for (int yi = 0; yi < H; yi++) {
        frag_t q[256];
        const float yf = pix_to_ndc(H - 1 - yi, H, W);
        for (int xi = 0; xi < W; xi++) {
            const float xf = pix_to_ndc(W - 1 - xi, W, H);
            int q_size = 0, q_max_idx = -1;
            float q_max_z = -1000.0f;
            for (int64_t f = 0; f < nc; f++) {
                frag_t fr;
                const float* bx = box + 4 * f;
                if (!(bx[0] <= xf && xf <= bx[1] && bx[2] <= yf && yf <= bx[3])) continue;
                if (eval_pixel_face(cf[f].v, xf, yf, blur_radius, sqrt_blur,
                                    perspective_correct, clip_bary, cull_backfaces, &fr))
                    kbuf_insert_entry(q, &q_size, &q_max_z, &q_max_idx, K, &fr, cf, f);
            }
            kbuf_sort(q, q_size);
            const int64_t base = ((int64_t)yi * W + xi) * K;
            for (int k = 0; k < K; k++) {
                if (k < q_size) {
                    pix_to_face[base + k] = q[k].face;
                    zbuf[base + k] = q[k].z;
                    dists[base + k] = q[k].sdist;
                    bary[(base + k) * 3 + 0] = q[k].b0;
                    bary[(base + k) * 3 + 1] = q[k].b1;
                    bary[(base + k) * 3 + 2] = q[k].b2;
                } else {
                    pix_to_face[base + k] = -1;
                    zbuf[base + k] = -1.0f;
                    dists[base + k] = -1.0f;
                    bary[(base + k) * 3 + 0] = -1.0f;
                    bary[(base + k) * 3 + 1] = -1.0f;
                    bary[(base + k) * 3 + 2] = -1.0f;
                }
            }
        }
    }
    free(box);
    free(cf);
    return 0;
}

/*
 * One render pass = what the reference obtains from renderer(mesh) [K=1] plus
 * sil_renderer(mesh) [K=K_sil] on the same mesh (pipelines.py:1546-1547), in a
 * single sweep: nearest fragment per pixel + the compact list of every fragment
 * kept by the K_sil-buffer (pixel, face, sub-triangle, signed dist), sorted by z per pixel.
 * The list is returned malloc'ed; free it with foho_oracle_free.
 * count[p] = number of kept fragments of pixel p.
 */
int foho_oracle_render_pass(const float* face_verts, int64_t F, int H, int W, float blur_radius,
                            int K_sil, int64_t* pix_to_face, float* zbuf, float* bary,
                            float* dists, int32_t* count, int8_t* sub, int64_t** pairs_out,
                            float** pair_dist_out, int64_t* n_pairs_out) {
    /* sub (H,W): which sub-triangle of a near-clipped face the nearest fragment comes from (-1: the face itself);
     * pairs are (pixel, face, sub) triples */
    if (K_sil < 1 || K_sil > 256) return -1;
    const float sqrt_blur = sqrtf(blur_radius);
    int64_t nc = 0;
    cface_t* cf = clip_faces(face_verts, F, 1, &nc);
    float* box = make_face_boxes(cf, nc, sqrt_blur);
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    if (nthreads > H) nthreads = H;
    int64_t** t_pairs = (int64_t**)calloc(nthreads, sizeof(int64_t*));
    float** t_dist = (float**)calloc(nthreads, sizeof(float*));
    int64_t* t_n = (int64_t*)calloc(nthreads, sizeof(int64_t));
    int64_t* t_cap = (int64_t*)calloc(nthreads, sizeof(int64_t));
#pragma omp parallel num_threads(nthreads)
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        const int r0 = (int)((int64_t)H * t / nthreads), r1 = (int)((int64_t)H * (t + 1) / nthreads);
        frag_t q[256];
        for (int yi = r0; yi < r1; yi++) {
            const float yf = pix_to_ndc(H - 1 - yi, H, W);
            for (int xi = 0; xi < W; xi++) {
                const float xf = pix_to_ndc(W - 1 - xi, W, H);
                int q_size = 0, q_max_idx = -1;
                float q_max_z = -1000.0f;
                for (int64_t f = 0; f < nc; f++) {
                    frag_t fr;
                    const float* bx = box + 4 * f;
                    if (!(bx[0] <= xf && xf <= bx[1] && bx[2] <= yf && yf <= bx[3])) continue;
                    if (eval_pixel_face(cf[f].v, xf, yf, blur_radius, sqrt_blur, 1, 1, 0, &fr))
                        kbuf_insert_entry(q, &q_size, &q_max_z, &q_max_idx, K_sil, &fr, cf, f);
                }
                kbuf_sort(q, q_size);
                const int64_t p = (int64_t)yi * W + xi;
                count[p] = q_size;
                sub[p] = (int8_t)((q_size > 0) ? q[0].sub : -1);
                if (q_size > 0) {
                    pix_to_face[p] = q[0].face;
                    zbuf[p] = q[0].z;
                    dists[p] = q[0].sdist;
                    bary[p * 3 + 0] = q[0].b0;
                    bary[p * 3 + 1] = q[0].b1;
                    bary[p * 3 + 2] = q[0].b2;
                } else {
                    pix_to_face[p] = -1;
                    zbuf[p] = -1.0f;
                    dists[p] = -1.0f;
                    bary[p * 3 + 0] = bary[p * 3 + 1] = bary[p * 3 + 2] = -1.0f;
                }
                if (t_n[t] + q_size > t_cap[t]) {
                    t_cap[t] = (t_cap[t] + q_size) * 2 + 1024;
                    t_pairs[t] = (int64_t*)realloc(t_pairs[t], sizeof(int64_t) * 3 * t_cap[t]);
                    t_dist[t] = (float*)realloc(t_dist[t], sizeof(float) * t_cap[t]);
                }
                for (int k = 0; k < q_size; k++) {
                    t_pairs[t][3 * t_n[t] + 0] = p;
                    t_pairs[t][3 * t_n[t] + 1] = q[k].face;
                    t_pairs[t][3 * t_n[t] + 2] = q[k].sub;
                    t_dist[t][t_n[t]] = q[k].sdist;
                    t_n[t]++;
                }
            }
        }
    }
    int64_t total = 0;
    for (int t = 0; t < nthreads; t++) total += t_n[t];
    int64_t* pairs = (int64_t*)malloc(sizeof(int64_t) * 3 * (total > 0 ? total : 1));
    float* pd = (float*)malloc(sizeof(float) * (total > 0 ? total : 1));
    int64_t o = 0;
    for (int t = 0; t < nthreads; t++) {
        if (t_n[t]) {
            memcpy(pairs + 3 * o, t_pairs[t], sizeof(int64_t) * 3 * t_n[t]);
            memcpy(pd + o, t_dist[t], sizeof(float) * t_n[t]);
        }
        o += t_n[t];
        free(t_pairs[t]);
        free(t_dist[t]);
    }
    free(box);
    free(cf);
    free(t_pairs);
    free(t_dist);
    free(t_n);
    free(t_cap);
    *pairs_out = pairs;
    *pair_dist_out = pd;
    *n_pairs_out = total;
    return 0;
}

void foho_oracle_free(void* p) { free(p); }

/* ------------------------------------------------------------------------- */
/* inside test: +z ray parity with an exact-once shared-edge rule              */
/* ------------------------------------------------------------------------- */

/* "left of the directed edge i->j".  The edge function is always evaluated
 * from the lower-index endpoint so the two faces sharing an edge see exactly
 * negated values.  A point exactly ON the edge line (e == 0) is classified as
 * the symbolically perturbed point p + (eps, eps^2) would be: the sign of
 * -(dy) eps + (dx) eps^2 for the directed edge (dx, dy).  The perturbed point is
 * in general position, so rays through edges AND vertices are counted once. */
static inline int left_of(const float* V, int32_t i, int32_t j, float px, float py, float* e_out) {
    float e;
    if (i < j) {
        const float ax = V[3 * i], ay = V[3 * i + 1], bx = V[3 * j], by = V[3 * j + 1];
        e = (bx - ax) * (py - ay) - (by - ay) * (px - ax);
    } else {
        const float ax = V[3 * j], ay = V[3 * j + 1], bx = V[3 * i], by = V[3 * i + 1];
        e = -((bx - ax) * (py - ay) - (by - ay) * (px - ax));
    }
    *e_out = e;
    if (e > 0.0f) return 1;
    if (e < 0.0f) return 0;
    {
        const float dx = V[3 * j] - V[3 * i], dy = V[3 * j + 1] - V[3 * i + 1];
        return (dy < 0.0f) || (dy == 0.0f && dx > 0.0f);
    }
}

/* Does the +z ray from (px,py,pz) cross face (ia,ib,ic)?  The xy bounding-box
 * test is part of the definition (not only an early-out): a face can only be
 * crossed by rays whose (px,py) lies inside its closed xy bounding box. */
static inline int ray_crosses(const float* V, int32_t ia, int32_t ib, int32_t ic, float px,
                              float py, float pz) {
    const float xa = V[3 * ia], xb = V[3 * ib], xc = V[3 * ic];
    if (px < fminf(fminf(xa, xb), xc) || px > fmaxf(fmaxf(xa, xb), xc)) return 0;
    const float ya = V[3 * ia + 1], yb = V[3 * ib + 1], yc = V[3 * ic + 1];
    if (py < fminf(fminf(ya, yb), yc) || py > fmaxf(fmaxf(ya, yb), yc)) return 0;
    float e0, e1, e2; /* weights of a, b, c */
    const int s0 = left_of(V, ib, ic, px, py, &e0);
    const int s1 = left_of(V, ic, ia, px, py, &e1);
    const int s2 = left_of(V, ia, ib, px, py, &e2);
    if (!((s0 && s1 && s2) || (!s0 && !s1 && !s2))) return 0;
    const float area = (e0 + e1) + e2;
    if (area == 0.0f) return 0;
    const float zh = ((e0 * V[3 * ia + 2] + e1 * V[3 * ib + 2]) + e2 * V[3 * ic + 2]) / area;
    return zh > pz;
}

/* inside[n] = 1 when pts[n] lies inside the closed mesh (odd crossing count). */
int foho_oracle_inside(const float* verts, int64_t V, const int32_t* faces, int64_t F,
                       const float* pts, int64_t N, uint8_t* inside) {
    (void)V;
    /* per-face xy boxes, only to skip ray_crosses() early (it re-tests the same box) */
    float* box = (float*)malloc(sizeof(float) * 4 * (F > 0 ? F : 1));
    for (int64_t f = 0; f < F; f++) {
        const float* a = verts + 3 * faces[3 * f];
        const float* b = verts + 3 * faces[3 * f + 1];
        const float* c = verts + 3 * faces[3 * f + 2];
        box[4 * f + 0] = fminf(fminf(a[0], b[0]), c[0]);
        box[4 * f + 1] = fmaxf(fmaxf(a[0], b[0]), c[0]);
        box[4 * f + 2] = fminf(fminf(a[1], b[1]), c[1]);
        box[4 * f + 3] = fmaxf(fmaxf(a[1], b[1]), c[1]);
    }
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        const float px = pts[3 * n], py = pts[3 * n + 1], pz = pts[3 * n + 2];
        int par = 0;
        for (int64_t f = 0; f < F; f++) {
            const float* bx = box + 4 * f;
            if (px < bx[0] || px > bx[1] || py < bx[2] || py > bx[3]) continue;
            par ^= ray_crosses(verts, faces[3 * f], faces[3 * f + 1], faces[3 * f + 2], px, py, pz);
        }
        inside[n] = (uint8_t)par;
    }
    free(box);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* exact point -> triangle squared distance (brute force over faces)           */
/* ------------------------------------------------------------------------- */

static inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static float point_tri_d2(const float* p, const float* a, const float* b, const float* c) {
    /* closest point on triangle (Ericson, Real-Time Collision Detection 5.1.5) */
    float ab[3], ac[3], ap[3], bp[3], cp[3], q[3];
    for (int k = 0; k < 3; k++) {
        ab[k] = b[k] - a[k];
        ac[k] = c[k] - a[k];
        ap[k] = p[k] - a[k];
    }
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0.0f && d2 <= 0.0f) {
        return dot3(ap, ap);
    }
    for (int k = 0; k < 3; k++) bp[k] = p[k] - b[k];
    const float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0.0f && d4 <= d3) {
        return dot3(bp, bp);
    }
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) {
        const float v = d1 / (d1 - d3);
        for (int k = 0; k < 3; k++) q[k] = p[k] - (a[k] + v * ab[k]);
        return dot3(q, q);
    }
    for (int k = 0; k < 3; k++) cp[k] = p[k] - c[k];
    const float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0.0f && d5 <= d6) {
        return dot3(cp, cp);
    }
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) {
        const float w = d2 / (d2 - d6);
        for (int k = 0; k < 3; k++) q[k] = p[k] - (a[k] + w * ac[k]);
        return dot3(q, q);
    }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int k = 0; k < 3; k++) q[k] = p[k] - (b[k] + w * (c[k] - b[k]));
        return dot3(q, q);
    }
    const float denom = 1.0f / (va + vb + vc);
    const float v = vb * denom, w = vc * denom;
    for (int k = 0; k < 3; k++) q[k] = p[k] - (a[k] + ab[k] * v + ac[k] * w);
    return dot3(q, q);
}

int foho_oracle_point_mesh_dist(const float* verts, int64_t V, const int32_t* faces, int64_t F,
                                const float* pts, int64_t N, float* d2_out, int64_t* face_out) {
    (void)V;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; n++) {
        float best = INFINITY;
        int64_t bf = -1;
        for (int64_t f = 0; f < F; f++) {
            const float d = point_tri_d2(pts + 3 * n, verts + 3 * faces[3 * f],
                                         verts + 3 * faces[3 * f + 1], verts + 3 * faces[3 * f + 2]);
            if (d < best) {
                best = d;
                bf = f;
            }
        }
        d2_out[n] = best;
        if (face_out) face_out[n] = bf;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* K=1 nearest neighbour, squared L2, ties -> lowest index                     */
/* ------------------------------------------------------------------------- */
int foho_oracle_knn1(const float* p1, int64_t N1, const float* p2, int64_t N2, float* d2_out,
                     int64_t* idx_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N1; i++) {
        float best = INFINITY;
        int64_t bi = -1;
        for (int64_t j = 0; j < N2; j++) {
            const float dx = p1[3 * i] - p2[3 * j], dy = p1[3 * i + 1] - p2[3 * j + 1],
                        dz = p1[3 * i + 2] - p2[3 * j + 2];
            const float d = dx * dx + dy * dy + dz * dz;
            if (d < best) {
                best = d;
                bi = j;
            }
        }
        d2_out[i] = best;
        idx_out[i] = bi;
    }
    return 0;
}

int foho_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
