"""Minimal mesh / array file IO at the edges of the hot path.

The reference reads `{idx}_hamer_aligned_mano.ply` with pytorch3d.io.load_ply (pipelines.py:1223), HaMeR's
`{idx}_hamer.obj` and Hunyuan's `{idx}_hoi_mesh.ply` with trimesh.load(process=False)
(src/foho/alignment/mesh_align.py:186-187), writes `{idx}_obj.ply` / `{idx}_hand.ply`
(src/foho/guidance/run.py:164-166) and the 4x4 `.npy` transform (mesh_align.py:207).  trimesh / pytorch3d are not
available on the MI355X image, so the two formats are parsed here: PLY (ascii, binary little/big endian;
vertex xyz + optional extra properties, triangular or polygonal faces, point clouds) and Wavefront OBJ (v / f).
"""
import os
import struct

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def _read_uniform_list_element(f, el, end):
    """Binary element whose list properties all have the same length in every row (the usual all-triangle face element):
    read it as one structured array.  Returns the (count, n) vertex-index array, False when the element has no vertex
    index list, or None (file position restored) when the rows are not uniform."""
    pos = f.tell()
    if el["count"] == 0:
        return False
    # list lengths of the first row
    fields, lens = [], []
    for q, p in enumerate(el["props"]):
        if p[0] == "list":
            cdt, idt = np.dtype(end + _PLY_TYPES[p[1]]), np.dtype(end + _PLY_TYPES[p[2]])
            raw = f.read(cdt.itemsize)
            if len(raw) < cdt.itemsize:
                f.seek(pos)
                return None
            n = int(np.frombuffer(raw, cdt)[0])
            f.seek(idt.itemsize * n, 1)
            fields += [(f"c{q}", cdt), (f"l{q}", idt, (n,))]
            lens.append((q, n, p[3]))
        else:
            dt = np.dtype(end + _PLY_TYPES[p[0]])
            f.seek(dt.itemsize, 1)
            fields.append((f"s{q}", dt))
    f.seek(pos)
    dt = np.dtype(fields)
    raw = f.read(dt.itemsize * el["count"])
    if len(raw) < dt.itemsize * el["count"]:
        f.seek(pos)
        return None
    data = np.frombuffer(raw, dtype=dt, count=el["count"])
    for q, n, _ in lens:
        if not np.all(data[f"c{q}"] == n):
            f.seek(pos)
            return None
    for q, n, name in lens:
        if name in ("vertex_indices", "vertex_index"):
            return data[f"l{q}"].reshape(el["count"], n)
    return False


def load_ply(path):
    """Returns (verts (V,3) float32, faces (F,3) int64).  Faces with more than 3 vertices are fan-triangulated;
    a file without a face element (point cloud) returns an empty (0,3) face array."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1]["props"].append(("list", tok[2], tok[3], tok[4]))
                else:
                    elements[-1]["props"].append((tok[1], tok[2]))
            elif tok[0] == "end_header":
                break
        verts, faces = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64)
        if fmt == "ascii":
            for el in elements:
                rows = [f.readline().split() for _ in range(el["count"])]
                if el["name"] == "vertex":
                    names = [p[1] for p in el["props"]]
                    ix = [names.index(c) for c in ("x", "y", "z")]
                    verts = np.array([[float(r[i]) for i in ix] for r in rows], np.float32).reshape(-1, 3)
                elif el["name"] == "face":
                    tri = []
                    for r in rows:
                        n = int(r[0])
                        idx = [int(x) for x in r[1:1 + n]]
                        tri += [[idx[0], idx[k], idx[k + 1]] for k in range(1, n - 1)]
                    faces = np.array(tri, np.int64).reshape(-1, 3)
            return verts, faces
        end = "<" if fmt == "binary_little_endian" else ">"
        for el in elements:
            has_list = any(p[0] == "list" for p in el["props"])
            if not has_list:
                dt = np.dtype([(p[1], end + _PLY_TYPES[p[0]]) for p in el["props"]])
                data = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt, count=el["count"])
                if el["name"] == "vertex":
                    verts = np.stack([data["x"], data["y"], data["z"]], 1).astype(np.float32)
            else:
                fast = _read_uniform_list_element(f, el, end)
                if fast is not None:
                    if el["name"] == "face" and fast is not False:
                        n = fast.shape[1]
                        faces = np.stack([np.stack([fast[:, 0], fast[:, k], fast[:, k + 1]], 1) for k in range(1, n - 1)],
                                         1).reshape(-1, 3).astype(np.int64) if n >= 3 else faces
                    continue
                tri = []
                for _ in range(el["count"]):
                    row_idx = None
                    for p in el["props"]:
                        if p[0] == "list":
                            cdt, idt = np.dtype(end + _PLY_TYPES[p[1]]), np.dtype(end + _PLY_TYPES[p[2]])
                            n = int(np.frombuffer(f.read(cdt.itemsize), cdt)[0])
                            idx = np.frombuffer(f.read(idt.itemsize * n), idt).astype(np.int64)
                            if p[3] in ("vertex_indices", "vertex_index"):
                                row_idx = idx
                        else:
                            f.read(np.dtype(_PLY_TYPES[p[0]]).itemsize)
                    if el["name"] == "face" and row_idx is not None:
                        tri += [[row_idx[0], row_idx[k], row_idx[k + 1]] for k in range(1, len(row_idx) - 1)]
                if el["name"] == "face":
                    faces = np.array(tri, np.int64).reshape(-1, 3)
        return verts, faces


def save_ply(path, verts, faces=None, binary=True):
    """Triangle mesh (or point cloud when faces is None/empty) as PLY, float32 vertices / int32 indices."""
    v = np.asarray(verts, np.float32).reshape(-1, 3)
    fc = np.zeros((0, 3), np.int32) if faces is None else np.asarray(faces, np.int32).reshape(-1, 3)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        hdr = ["ply", "format binary_little_endian 1.0" if binary else "format ascii 1.0", f"element vertex {len(v)}",
               "property float x", "property float y", "property float z"]
        if len(fc):
            hdr += [f"element face {len(fc)}", "property list uchar int vertex_indices"]
        hdr.append("end_header")
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        if binary:
            f.write(v.astype("<f4").tobytes())
            if len(fc):
                rec = np.empty(len(fc), dtype=[("n", "u1"), ("i", "<i4", 3)])
                rec["n"], rec["i"] = 3, fc
                f.write(rec.tobytes())
        else:
            for p in v:
                f.write(f"{p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n".encode())
            for t in fc:
                f.write(f"3 {t[0]} {t[1]} {t[2]}\n".encode())


def load_obj(path):
    """Wavefront OBJ: 'v x y z' and 'f a[/..] b[/..] c[/..] ...' (1-based, negative = relative)."""
    vs, fs = [], []
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        for line in f:
            if line.startswith("v "):
                t = line.split()
                vs.append([float(t[1]), float(t[2]), float(t[3])])
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(vs) + i)
                fs += [[idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1)]
    return np.array(vs, np.float32).reshape(-1, 3), np.array(fs, np.int64).reshape(-1, 3)


def save_obj(path, verts, faces):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        for p in np.asarray(verts):
            f.write(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")
        for t in np.asarray(faces):
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")


def load_mesh(path):
    """Dispatch on the extension like trimesh.load(process=False) / pytorch3d's IO for the formats the path uses
    (.glb: the MoGe image mesh, PL:1247-1250)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".ply":
        return load_ply(path)
    if ext == ".obj":
        return load_obj(path)
    if ext == ".glb":
        from .inputs import load_glb
        return load_glb(path)
    raise ValueError(f"unsupported mesh format '{ext}' ({path}); supported: .ply, .obj, .glb")
