"""`python3 -m foho.alignment.h2m` -- align every Hunyuan HOI mesh to its MoGe geometry and write the 4x4
Hunyuan->MoGe transform (reference src/foho/alignment/h2m.py:12-68; same flags, same file names, same ICP
settings: 50/1000/5000 coarse, 100/5000/10000 fine, 20 % outliers, scale in [0.7, 3.0])."""
import argparse
import glob
import os

from foho.alignment.mesh_align import align_meshes_impl

ICP_SETTINGS = dict(fixed_scale=False, outliers=0.2, test_rotations=False, test_reflections=False, on_surface=False,
                    iterations_coarse=50, count_source_coarse=1000, count_target_coarse=5000, iterations_fine=100,
                    count_source_fine=5000, count_target_fine=10000, min_scale=0.7, max_scale=3.0, plot=False)


def each_source(directory: str, pattern: str, what: str):
    """[(path, image index, file stem)] of the source meshes of a stage; the index is the text before the first "_"."""
    found = sorted(glob.glob(os.path.join(directory, pattern)))
    if not found:
        print(f"No {what} found in {directory}")
    return [(p, os.path.basename(p).split("_")[0], os.path.splitext(os.path.basename(p))[0]) for p in found]


def build_parser(*flags) -> argparse.ArgumentParser:
    """The stages take only required --<name> directory flags (h2m.py:57-68, mano.py:46-57)."""
    parser = argparse.ArgumentParser()
    for flag in flags:
        parser.add_argument(f"--{flag}", required=True)
    return parser


def cli(run_fn, *flags, argv=None) -> None:
    run_fn(**vars(build_parser(*flags).parse_args(argv)))


def pick_moge_target(moge_dir: str):
    """h2m.py:24-34: mesh.ply, else pointcloud.ply, else mesh.glb (read by followmyhold_amd.inputs.load_glb)."""
    for name in ("mesh.ply", "pointcloud.ply", "mesh.glb"):
        p = os.path.join(moge_dir, name)
        if os.path.isfile(p):
            return p
    return None


def run(hunyuan_mesh_dir: str, moge_out_dir: str, h2m_rt_dir: str) -> None:
    sources = each_source(hunyuan_mesh_dir, "*.ply", "Hunyuan HOI meshes")
    if sources:
        os.makedirs(h2m_rt_dir, exist_ok=True)
    for path, index, stem in sources:
        moge_dir = os.path.join(moge_out_dir, f"{index}_cropped_hoi")
        target = pick_moge_target(moge_dir)
        if target is None:
            print(f"No MoGe mesh found for {index} in {moge_dir}. Skipping.")
            continue
        # np.save appends ".npy": the guidance stage reads {index}_hoi_mesh.npy (run.py:216)
        align_meshes_impl(source_mesh_path=path, target_mesh_path=target, transform_path=os.path.join(h2m_rt_dir, stem),
                          transformed_mesh_path=None, **ICP_SETTINGS)


FLAGS = ("hunyuan_mesh_dir", "moge_out_dir", "h2m_rt_dir")


def main() -> None:
    cli(run, *FLAGS)


if __name__ == "__main__":
    main()
