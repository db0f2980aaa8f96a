"""`python3 -m foho.alignment.mano` -- align every HaMeR hand mesh (.obj) to the Hunyuan HOI mesh and write
`{name}_aligned_mano.ply` (reference src/foho/alignment/mano.py:12-57; same flags, names and ICP settings)."""
import os

from foho.alignment.h2m import ICP_SETTINGS, cli, each_source
from foho.alignment.mesh_align import align_meshes_impl


def run(hamer_out_dir: str, hunyuan_mesh_dir: str, aligned_mano_dir: str) -> None:
    sources = each_source(hamer_out_dir, "*.obj", "HaMeR meshes")
    if sources:
        os.makedirs(aligned_mano_dir, exist_ok=True)
    for path, index, stem in sources:
        align_meshes_impl(source_mesh_path=path, target_mesh_path=os.path.join(hunyuan_mesh_dir, f"{index}_hoi_mesh.ply"),
                          transform_path=None, transformed_mesh_path=os.path.join(aligned_mano_dir, f"{stem}_aligned_mano.ply"),
                          **ICP_SETTINGS)


FLAGS = ("hamer_out_dir", "hunyuan_mesh_dir", "aligned_mano_dir")


def main() -> None:
    cli(run, *FLAGS)


if __name__ == "__main__":
    main()
