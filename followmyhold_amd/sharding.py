"""Image sharding across the GPUs of one node and the end-of-batch metrics all-reduce.

The reference scales out with SLURM job arrays over a JSON list of filename chunks
(src/foho/guidance/run.py:178-185): every image's guidance loop is independent.  The MI355X counterpart is one
process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), rank r owning images r, r+R, r+2R, ...
of the sorted list, and a single all-reduce(SUM) of a small fp64 metrics vector when the batch is done
(SURVEY.md 8(e)).  There is no collective on the data path.
"""
import json
import os
from typing import List, Optional

import torch

METRIC_NAMES = ["n_images", "n_steps", "sum_total_loss", "sum_intersection", "sum_contact", "sum_kps", "sum_edge",
                "sum_normal_hand", "sum_disp_hand", "sum_normal_hoi", "sum_disp_hoi", "sum_sil_hoi", "sum_wall_ms",
                "n_nan", "n_flagged", "n_failed"]
IDX = {n: i for i, n in enumerate(METRIC_NAMES)}


def shard_images(items: List[str], rank: int, world_size: int) -> List[str]:
    """Round-robin partition of the sorted image list: rank r gets items[r::world_size]."""
    return list(items)[rank::world_size]


def load_task_list(task_list_file: Optional[str], cropped_obj_img_dir: str, rank: Optional[int] = None,
                   world_size: Optional[int] = None) -> List[str]:
    """Reference semantics (run.py:178-185): a JSON list of chunks indexed by SLURM_ARRAY_TASK_ID when the file
    exists, else the sorted directory listing -- then, when running under torch.distributed, this rank's
    round-robin share of it."""
    if task_list_file and os.path.exists(task_list_file):
        with open(task_list_file, "r", encoding="utf-8") as f:
            chunks = json.load(f)
        items = chunks[int(os.environ.get("SLURM_ARRAY_TASK_ID", 0))]
    else:
        items = sorted(os.listdir(cropped_obj_img_dir))
    if rank is None:
        rank = int(os.environ.get("RANK", 0))
    if world_size is None:
        world_size = int(os.environ.get("WORLD_SIZE", 1))
    return shard_images(items, rank, world_size) if world_size > 1 else list(items)


_TERM_COLS = [0, 1, 2, 3, 7, 8, 9, 11, 12, 13]   # total, intersection, contact, kps, edge, normal0, disp0, normal1, disp1, sil1


def image_metrics(losses_row, flags: int, n_steps: int):
    """One image's contribution to the metrics vector (list of floats, METRIC_NAMES order) from its row of
    GuidanceBatch.losses and its flag word; wall time and failures are tallied by the driver."""
    import math
    l = [float(x) for x in losses_row]
    v = [1.0, float(n_steps)] + [0.0 if math.isnan(l[c]) else l[c] for c in _TERM_COLS]
    return v + [0.0, float(bool(flags & 1)), float(bool(flags & 6)), 0.0]


def local_metrics(gb, n_steps: int, wall_ms: float) -> torch.Tensor:
    """This rank's contribution: sums over its images of the last step's loss terms (device -> fp64 vector)."""
    l = gb.losses.detach().double()
    fl = gb.flags.detach()
    v = [float(gb.B), float(n_steps) * gb.B]
    for c in _TERM_COLS:
        v.append(float(torch.nan_to_num(l[:, c]).sum().item()))
    # n_failed (images whose processing raised) is tallied by the driver, not here
    v += [wall_ms, float((fl & 1).ne(0).sum().item()), float((fl & 6).ne(0).sum().item()), 0.0]
    return torch.tensor(v, dtype=torch.float64, device=gb.losses.device)


def all_reduce_metrics(vec: torch.Tensor, dist=None) -> torch.Tensor:
    """One all-reduce(SUM) over the node (RCCL on GPUs, gloo in CPU tests)."""
    if dist is None:
        import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return vec
