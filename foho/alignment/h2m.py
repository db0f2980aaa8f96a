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


def pick_moge_target(moge_dir: str):
    """h2m.py:24-34: mesh.ply, else pointcloud.ply, else mesh.glb (read by followmyhold_amd.inputs.load_glb)."""
    for name in ("mesh.ply", "pointcloud.ply", "mesh.glb"):
        p = os.path.join(moge_dir, name)
        if os.path.isfile(p):
            return p
    return None


def run(hunyuan_mesh_dir: str, moge_out_dir: str, h2m_rt_dir: str) -> None:
    meshes = sorted(glob.glob(os.path.join(hunyuan_mesh_dir, "*.ply")))
    if not meshes:
        print(f"No Hunyuan HOI meshes found in {hunyuan_mesh_dir}")
        return
    os.makedirs(h2m_rt_dir, exist_ok=True)
    for mesh_path in meshes:
        base_name = os.path.basename(mesh_path)
        i = base_name.split("_")[0]
        j = os.path.splitext(base_name)[0]
        moge_dir = os.path.join(moge_out_dir, f"{i}_cropped_hoi")
        target_mesh = pick_moge_target(moge_dir)
        if target_mesh is None:
            print(f"No MoGe mesh found for {i} in {moge_dir}. Skipping.")
            continue
        # np.save appends ".npy": the guidance stage reads {i}_hoi_mesh.npy (run.py:216)
        align_meshes_impl(source_mesh_path=mesh_path, target_mesh_path=target_mesh, transform_path=os.path.join(h2m_rt_dir, j),
                          transformed_mesh_path=None, **ICP_SETTINGS)


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--hunyuan_mesh_dir", required=True)
    parser.add_argument("--moge_out_dir", required=True)
    parser.add_argument("--h2m_rt_dir", required=True)
    a = parser.parse_args()
    run(hunyuan_mesh_dir=a.hunyuan_mesh_dir, moge_out_dir=a.moge_out_dir, h2m_rt_dir=a.h2m_rt_dir)


if __name__ == "__main__":
    main()
