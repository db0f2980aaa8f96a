"""ctypes binding of libfoho_hip.so (C ABI declared in include/foho_hip.h).

The HIP library is the product path: importing this module never falls back to a CPU
implementation -- `lib()` raises if the shared object is missing and every wrapper raises
FohoError on a non-zero status.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("FOHO_HIP_SO") or os.path.join(_HERE, "libfoho_hip.so")   # $FOHO_HIP_SO: another build of the same library (development)
_lib = None

c_f = ctypes.c_float
c_i = ctypes.c_int32
vp = ctypes.c_void_p


class FohoError(RuntimeError):
    pass


class FohoImage(ctypes.Structure):
    _fields_ = [("v_off", c_i), ("Vh", c_i), ("Vo", c_i), ("f_off", c_i), ("Fh", c_i), ("Fo", c_i),
                ("n_edges", c_i), ("jcols", c_i), ("k00", c_f), ("k11", c_f), ("cam_R", c_f * 9),
                ("cam_T", c_f * 3), ("znear", c_f), ("zfar", c_f), ("T_h2m", c_f * 12)]


class FohoDims(ctypes.Structure):
    _fields_ = [("B", c_i), ("H", c_i), ("W", c_i), ("Vtot", c_i), ("Ftot", c_i), ("Vmax", c_i), ("Fmax", c_i),
                ("Vh_max", c_i), ("Vo_max", c_i), ("grid_res", c_i), ("frac_cap", c_i), ("n_renders", c_i),
                ("Fh_max", c_i), ("Fo_max", c_i), ("gbuf_f16", c_i)]


class FohoRenderCfg(ctypes.Structure):
    _fields_ = [("face_set", c_i), ("normal_mask", c_i), ("disp_mask", c_i), ("sil_mask", c_i),
                ("w_normal", c_f), ("w_disp", c_f), ("w_sil", c_f)]


class FohoStepCfg(ctypes.Structure):
    _fields_ = [("render", FohoRenderCfg * 2), ("w_kps", c_f), ("w_trans_hand", c_f), ("w_trans_obj", c_f),
                ("w_verts_obj", c_f), ("w_edge", c_f), ("w_contact", c_f), ("contact_margin", c_f),
                ("use_intersection", c_i), ("w_int_near", c_f), ("w_int_far", c_f), ("int_gate", c_f),
                ("int_gate_step_ok", c_i), ("sigma", c_f), ("gamma", c_f), ("blur_radius", c_f),
                ("lr", c_f * 16), ("beta1", c_f), ("beta2", c_f), ("eps", c_f), ("weight_decay", c_f),
                ("do_update", c_i), ("world_space_input", c_i), ("deferred_update", c_i), ("n_active_renders", c_i),
                ("listed_cap", c_i)]


class FohoStepDesc(ctypes.Structure):
    _fields_ = [("dims", FohoDims), ("images", vp), ("verts_in", vp), ("faces", vp), ("inc_off", vp),
                ("inc_fc", vp), ("nbr_off", vp), ("nbr_idx", vp), ("J_regressor", vp), ("tgt_normal", vp),
                ("tgt_disp", vp), ("mask", vp), ("kps_2d", vp), ("params", vp), ("adam_m", vp), ("adam_v", vp),
                ("adam_t", vp), ("losses", vp), ("grad_params", vp), ("grad_verts_in", vp), ("flags", vp),
                ("workspace", vp), ("workspace_bytes", ctypes.c_size_t), ("hand_order_valid", c_i),
                ("hand_faces_per_block", c_i)]


# enums of include/foho_hip.h
FACES_HAND, FACES_OBJ, FACES_ALL = 0, 1, 2
MASK_NONE, MASK_HAND, MASK_OBJ, MASK_HOI = 0, 1, 2, 3
STAGE_VERTEX, STAGE_RASTER, STAGE_LOSS, STAGE_BACKWARD, STAGE_INSIDE, STAGE_FINAL, STAGE_BBOX = 1, 2, 4, 8, 16, 32, 64
STAGE_STEP, STAGE_ALL, STAGE_TARGETS = 63, 127, 256
N_LOSS = 24
LOSS_NAMES = ["total", "intersection", "contact", "kps", "trans_hand", "trans_obj", "verts_obj", "edge", "normal0",
              "disp0", "sil0", "normal1", "disp1", "sil1", "n_intersect", "w_int", "mean_d2"]
WS_REGIONS = ["world", "ndc", "vn", "p2f", "zbuf", "sdist", "prod", "knn_idx", "knn_d2", "gworld", "frac_count",
              "stats", "parity", "frag_count", "seg_count", "hand_order"]
N_KERNELS = 10
ABI_VERSION = 105     # 105 = foho_vae_fwd / _bwd, foho_geo_weights.flags, foho_geo_gemm variant bits, FOHO_API visibility; 104 = foho_geo_weights.ln_{q,kv,2}_eps, foho_geo_decode_bwd_rows, foho_geo_prepare_queries / _decode_fwd_cached; include/foho_hip.h: 102 = foho_step_cfg.listed_cap, foho_step_desc.hand_order_valid, foho_abi_sizes; 103 = foho_geo_weights.q_norm / k_norm, foho_geo_abi_size


def build(force=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir)] + [os.path.join(_HERE, "..", "include", "foho_hip.h")]
    if force or not os.path.exists(SO_PATH) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", src_dir, "-s"])
    return SO_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise FohoError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
        L = ctypes.CDLL(SO_PATH)
        L.foho_last_error.restype = ctypes.c_char_p
        L.foho_version.restype = ctypes.c_int
        sizes = (ctypes.c_int64 * 5)()
        L.foho_abi_sizes.restype = ctypes.c_int
        ver = L.foho_abi_sizes(sizes)
        mine = [ctypes.sizeof(t) for t in (FohoImage, FohoDims, FohoRenderCfg, FohoStepCfg, FohoStepDesc)]
        if list(sizes) != mine or ver < ABI_VERSION:
            raise FohoError(f"{SO_PATH} is ABI version {ver} with struct sizes {list(sizes)}; this binding is version "
                            f"{ABI_VERSION} with {mine}: rebuild the library (python -c 'import __graft_entry__ as g; g.build()')")
        L.foho_step_workspace_bytes.restype = ctypes.c_size_t
        L.foho_step_workspace_bytes.argtypes = [ctypes.POINTER(FohoDims)]
        L.foho_step_workspace_region.restype = ctypes.c_int64
        L.foho_step_workspace_region.argtypes = [ctypes.POINTER(FohoDims), ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
        L.foho_step_run.restype = ctypes.c_int
        L.foho_step_run.argtypes = [ctypes.POINTER(FohoStepDesc), ctypes.POINTER(FohoStepCfg), ctypes.c_int, vp]
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        raise FohoError(f"{what} failed ({status}): {lib().foho_last_error().decode()}")
