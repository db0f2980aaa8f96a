"""`python3 -m foho.alignment.mano` -- align every HaMeR hand mesh (.obj) to the Hunyuan HOI mesh and write
`{name}_aligned_mano.ply` (reference src/foho/alignment/mano.py:12-57; same flags, names and ICP settings)."""
import argparse
import glob
import os

from foho.alignment.h2m import ICP_SETTINGS
from foho.alignment.mesh_align import align_meshes_impl


def run(hamer_out_dir: str, hunyuan_mesh_dir: str, aligned_mano_dir: str) -> None:
    meshes = sorted(glob.glob(os.path.join(hamer_out_dir, "*.obj")))
    if not meshes:
        print(f"No HaMeR meshes found in {hamer_out_dir}")
        return
    os.makedirs(aligned_mano_dir, exist_ok=True)
    for mesh_path in meshes:
        base_name = os.path.basename(mesh_path)
        i = base_name.split("_")[0]
        j = os.path.splitext(base_name)[0]
        target_mesh = os.path.join(hunyuan_mesh_dir, f"{i}_hoi_mesh.ply")
        out_path = os.path.join(aligned_mano_dir, f"{j}_aligned_mano.ply")
        align_meshes_impl(source_mesh_path=mesh_path, target_mesh_path=target_mesh, transform_path=None,
                          transformed_mesh_path=out_path, **ICP_SETTINGS)


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--hamer_out_dir", required=True)
    parser.add_argument("--hunyuan_mesh_dir", required=True)
    parser.add_argument("--aligned_mano_dir", required=True)
    a = parser.parse_args()
    run(hamer_out_dir=a.hamer_out_dir, hunyuan_mesh_dir=a.hunyuan_mesh_dir, aligned_mano_dir=a.aligned_mano_dir)


if __name__ == "__main__":
    main()
