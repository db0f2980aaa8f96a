"""ctypes access to oracle/_build/libfoho_oracle.so (C restatement; test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfoho_oracle.so")
_lib = None


def build(force=False):
    """Compile the C oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "foho_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def set_z_clip(z=0.01 * 0.5):
    """Near plane of the restated rasteriser (default: znear / 2 of the path's camera); -inf-like values disable it."""
    lib().foho_oracle_set_z_clip(ctypes.c_float(np.float32(z)))


def count_near_clipped(face_verts):
    """Faces that straddle the near plane: clip_faces splits them into one or two sub-triangles."""
    fv = np.ascontiguousarray(face_verts, dtype=np.float32).reshape(-1, 9)
    f = lib().foho_oracle_count_near_clipped
    f.restype = ctypes.c_int64
    return int(f(_p(fv, ctypes.c_float), ctypes.c_int64(fv.shape[0])))


def get_z_clip():
    f = lib().foho_oracle_get_z_clip
    f.restype = ctypes.c_float
    return float(f())


def set_threads(n):
    return int(lib().foho_oracle_set_threads(int(n)))


def rasterize(face_verts, H, W, blur_radius, K=1, perspective_correct=True, clip_bary=True,
              cull_backfaces=False):
    """pytorch3d rasterize_meshes (naive) restatement.  face_verts (F,3,3) float32 NDC-xy + view z.
    Returns pix_to_face (H,W,K) int64, zbuf (H,W,K), bary (H,W,K,3), dists (H,W,K)."""
    fv = np.ascontiguousarray(face_verts, dtype=np.float32).reshape(-1, 9)
    F = fv.shape[0]
    p2f = np.empty((H, W, K), np.int64)
    zb = np.empty((H, W, K), np.float32)
    ba = np.empty((H, W, K, 3), np.float32)
    di = np.empty((H, W, K), np.float32)
    rc = lib().foho_oracle_rasterize(
        _p(fv, ctypes.c_float), ctypes.c_int64(F), H, W, ctypes.c_float(blur_radius), K,
        int(perspective_correct), int(clip_bary), int(cull_backfaces),
        _p(p2f, ctypes.c_int64), _p(zb, ctypes.c_float), _p(ba, ctypes.c_float), _p(di, ctypes.c_float))
    assert rc == 0
    return p2f, zb, ba, di


def render_pass(face_verts, H, W, blur_radius, K_sil=100):
    """Nearest fragment per pixel + compact list of all K_sil-buffer fragments.
    Returns dict(pix_to_face (H,W) int64, zbuf, bary (H,W,3), dists, count (H,W) int32, sub (H,W) int8 [sub-triangle of a
    near-clipped face the nearest fragment belongs to, -1 = the face itself], pairs (n,3) int64 [pixel, face, sub],
    pair_dist (n,) float32)."""
    fv = np.ascontiguousarray(face_verts, dtype=np.float32).reshape(-1, 9)
    F = fv.shape[0]
    p2f = np.empty((H, W), np.int64)
    zb = np.empty((H, W), np.float32)
    ba = np.empty((H, W, 3), np.float32)
    di = np.empty((H, W), np.float32)
    cnt = np.empty((H, W), np.int32)
    sub = np.empty((H, W), np.int8)
    pp = ctypes.POINTER(ctypes.c_int64)()
    pd = ctypes.POINTER(ctypes.c_float)()
    n = ctypes.c_int64(0)
    L = lib()
    rc = L.foho_oracle_render_pass(
        _p(fv, ctypes.c_float), ctypes.c_int64(F), H, W, ctypes.c_float(blur_radius), K_sil,
        _p(p2f, ctypes.c_int64), _p(zb, ctypes.c_float), _p(ba, ctypes.c_float), _p(di, ctypes.c_float),
        _p(cnt, ctypes.c_int32), _p(sub, ctypes.c_int8), ctypes.byref(pp), ctypes.byref(pd), ctypes.byref(n))
    assert rc == 0
    n = n.value
    if n > 0:
        pairs = np.ctypeslib.as_array(pp, shape=(n, 3)).copy()
        pdist = np.ctypeslib.as_array(pd, shape=(n,)).copy()
    else:
        pairs = np.zeros((0, 3), np.int64)
        pdist = np.zeros((0,), np.float32)
    L.foho_oracle_free.argtypes = [ctypes.c_void_p]
    L.foho_oracle_free(ctypes.cast(pp, ctypes.c_void_p))
    L.foho_oracle_free(ctypes.cast(pd, ctypes.c_void_p))
    return dict(pix_to_face=p2f, zbuf=zb, bary=ba, dists=di, count=cnt, sub=sub, pairs=pairs, pair_dist=pdist)


def inside(verts, faces, pts):
    v = np.ascontiguousarray(verts, np.float32)
    f = np.ascontiguousarray(faces, np.int32)
    p = np.ascontiguousarray(pts, np.float32)
    out = np.empty((p.shape[0],), np.uint8)
    rc = lib().foho_oracle_inside(_p(v, ctypes.c_float), ctypes.c_int64(v.shape[0]),
                                  _p(f, ctypes.c_int32), ctypes.c_int64(f.shape[0]),
                                  _p(p, ctypes.c_float), ctypes.c_int64(p.shape[0]),
                                  _p(out, ctypes.c_uint8))
    assert rc == 0
    return out.astype(bool)


def point_mesh_dist(verts, faces, pts):
    v = np.ascontiguousarray(verts, np.float32)
    f = np.ascontiguousarray(faces, np.int32)
    p = np.ascontiguousarray(pts, np.float32)
    d2 = np.empty((p.shape[0],), np.float32)
    fi = np.empty((p.shape[0],), np.int64)
    rc = lib().foho_oracle_point_mesh_dist(_p(v, ctypes.c_float), ctypes.c_int64(v.shape[0]),
                                           _p(f, ctypes.c_int32), ctypes.c_int64(f.shape[0]),
                                           _p(p, ctypes.c_float), ctypes.c_int64(p.shape[0]),
                                           _p(d2, ctypes.c_float), _p(fi, ctypes.c_int64))
    assert rc == 0
    return d2, fi


def knn1(p1, p2):
    a = np.ascontiguousarray(p1, np.float32)
    b = np.ascontiguousarray(p2, np.float32)
    d2 = np.empty((a.shape[0],), np.float32)
    idx = np.empty((a.shape[0],), np.int64)
    rc = lib().foho_oracle_knn1(_p(a, ctypes.c_float), ctypes.c_int64(a.shape[0]),
                                _p(b, ctypes.c_float), ctypes.c_int64(b.shape[0]),
                                _p(d2, ctypes.c_float), _p(idx, ctypes.c_int64))
    assert rc == 0
    return d2, idx
