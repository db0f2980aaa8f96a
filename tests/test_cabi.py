"""The C-ABI library loads and exports every symbol that include/*.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not fn.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)            # strip comments
        src = re.sub(r"//[^\n]*", "", src)
        src = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
        src = re.sub(r"typedef\s+enum\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
        src = re.sub(r"enum\s*\{.*?\}\s*;", "", src, flags=re.S)
        for m in re.finditer(r"\b(foho_\w+)\s*\(", src):
            names.append(m.group(1))
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    from followmyhold_amd import _lib
    _lib.build()
    return ctypes.CDLL(_lib.SO_PATH)


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ["foho_step_run", "foho_step_workspace_bytes", "foho_step_workspace_region", "foho_last_error",
                 "foho_version", "foho_step_run_profiled", "foho_geo_decode_fwd", "foho_vae_fwd", "foho_vae_bwd", "foho_sdpa_fwd", "foho_icp_run"]:
        assert must in names
    assert len(names) == 58, len(names)


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported by libfoho_hip.so: {missing}"


def test_nothing_but_the_declared_entry_points_is_exported():
    """-fvisibility=hidden + FOHO_API on the declarations: `nm -D` shows the header's functions and nothing else (no kernel stubs, no helpers)."""
    import subprocess
    from followmyhold_amd import _lib
    _lib.build()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.SO_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] in ("T", "t", "W", "V", "B", "D"))
    exported = [n for n in exported if not n.startswith(("_init", "_fini", "__bss_start", "_edata", "_end", "__hip_"))]
    assert exported == declared_functions(), sorted(set(exported) ^ set(declared_functions()))


def test_host_only_queries_work_without_a_gpu(lib):
    from followmyhold_amd import _lib as L
    lib.foho_version.restype = ctypes.c_int
    assert lib.foho_version() >= 100
    d = L.FohoDims()
    d.B, d.H, d.W, d.Vtot, d.Ftot, d.Vmax, d.Fmax, d.Vh_max, d.Vo_max = 2, 512, 512, 22040, 44064, 11020, 22032, 778, 10242
    d.grid_res, d.frac_cap, d.n_renders = 64, 1 << 18, 2
    lib.foho_step_workspace_bytes.restype = ctypes.c_size_t
    n = lib.foho_step_workspace_bytes(ctypes.byref(d))
    assert 10_000_000 < n < 2_000_000_000
    lib.foho_step_workspace_region.restype = ctypes.c_int64
    nb = ctypes.c_int64(0)
    off = lib.foho_step_workspace_region(ctypes.byref(d), L.WS_REGIONS.index("p2f"), ctypes.byref(nb))
    assert off >= 0 and nb.value == 2 * 2 * 512 * 512 * 4 and off + nb.value <= n
    assert lib.foho_step_workspace_region(ctypes.byref(d), 999, ctypes.byref(nb)) == -1
    # argument validation happens before any launch
    lib.foho_step_run.restype = ctypes.c_int
    assert lib.foho_step_run(None, None, 0, None) == -1
    lib.foho_last_error.restype = ctypes.c_char_p
    assert b"null" in lib.foho_last_error()
    # size queries and argument checks of the other entry points
    for fn, args, lo in [("foho_flexi_workspace_bytes", (64,), 1_000_000), ("foho_topology_workspace_bytes", (11020,), 50_000),
                         ("foho_raster_workspace_bytes", (11020, 22032, 512, 512), 1_000_000), ("foho_icp_workspace_bytes", (5000, 10000), 1)]:
        f = getattr(lib, fn)
        f.restype = ctypes.c_size_t
        assert f(*args) >= lo, fn
    assert lib.foho_flexi_workspace_bytes(0) == 0 and lib.foho_topology_workspace_bytes(0) == 0
    lib.foho_flexi_fwd.restype = ctypes.c_int
    assert lib.foho_flexi_fwd(None, None, 64, None, 0, None, 0, None, None, None, ctypes.c_size_t(0), None) == -1
    assert b"foho_flexi_fwd" in lib.foho_last_error()
    lib.foho_topology_tables.restype = ctypes.c_int
    assert lib.foho_topology_tables(None, 10, 10, None, None, None, None, None, None, ctypes.c_size_t(0), None) == -1


def test_struct_layouts_match_the_header(lib):
    """ctypes mirrors must have the C sizes (4-byte fields, 8-byte pointers, natural alignment)."""
    from followmyhold_amd import _lib as L
    assert ctypes.sizeof(L.FohoImage) == 8 * 4 + 2 * 4 + 9 * 4 + 3 * 4 + 2 * 4 + 12 * 4
    assert ctypes.sizeof(L.FohoDims) == 15 * 4
    assert ctypes.sizeof(L.FohoRenderCfg) == 7 * 4
    assert ctypes.sizeof(L.FohoStepCfg) == 2 * 28 + 7 * 4 + 4 + 3 * 4 + 4 + 3 * 4 + 16 * 4 + 4 * 4 + 4 + 4 + 4 + 4 + 4
    assert ctypes.sizeof(L.FohoStepDesc) == 64 + 21 * 8 + 8 + 8          # hand_order_valid + hand_faces_per_block
    # ... and the library this binding loads was built from the same layout (the check _lib.lib() makes at load time)
    sizes = (ctypes.c_int64 * 5)()
    lib.foho_abi_sizes.restype = ctypes.c_int
    assert lib.foho_abi_sizes(sizes) == lib.foho_version() == L.ABI_VERSION
    assert list(sizes) == [ctypes.sizeof(t) for t in (L.FohoImage, L.FohoDims, L.FohoRenderCfg, L.FohoStepCfg, L.FohoStepDesc)]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from followmyhold_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.FohoError):
        L.lib()
