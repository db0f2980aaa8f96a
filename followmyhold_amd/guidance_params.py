"""Per-phase optimiser parameter groups -- host mirror of the reference's third_party/utilz/code_utils.py:3-83.

Same signature, same return tuple, same error for an unknown phase.  A "fresh leaf" is
`tensor.clone().detach().requires_grad_(True)`; tensors that a phase does not optimise are handed back
untouched (the very same object), which callers rely on when they `.detach().clone()` everything afterwards
(pipelines.py:1604-1610).
"""
from typing import Dict, List, Optional, Tuple

import torch

# which tensors each phase turns into fresh leaves, in optimiser-group order: (name, lr table, lr key)
_PHASES = {
    1: [("scale_hand", "phase1_hand_lrs", "scale"), ("trans_hand", "phase1_hand_lrs", "trans"),
        ("rotation_hand", "phase1_hand_lrs", "rot")],
    1.5: [("scale_obj", "obj_2half_lrs", "scale"), ("trans_obj", "obj_2half_lrs", "trans"),
          ("rotation_obj", "obj_2half_lrs", "rot"), ("noise_pred_obj", "noise_obj_lr1", None)],
    2: [("scale_hand", "phase2_hand_lrs", "scale"), ("trans_hand", "phase2_hand_lrs", "trans"),
        ("rotation_hand", "phase2_hand_lrs", "rot"), ("scale_obj", "obj_lrs", "scale"),
        ("trans_obj", "obj_lrs", "trans"), ("rotation_obj", "obj_lrs", "rot"),
        ("noise_pred_obj", "noise_obj_lr2", None)],
}


def get_guidance_params(phase, noise_pred_obj, scale_hand, trans_hand, rotation_hand, device, *, phase1_hand_lrs,
                        phase2_hand_lrs, noise_obj_lr1, noise_obj_lr2, obj_lrs, obj_2half_lrs, obj_latent_lr=None,
                        scale_obj=None, trans_obj=None, rotation_obj=None):
    """Return (param_groups, noise_pred_obj, scale_hand, trans_hand, rotation_hand, scale_obj, trans_obj,
    rotation_obj) for phase 1 (hand only), 1.5 (object only) or 2 (joint)."""
    if phase not in _PHASES:
        raise ValueError(f"Unknown phase {phase}. Expected 'hand_only (1)', 'obj-only (1.5)' or 'joint_hand_obj (2)'.")
    lr_tables = dict(phase1_hand_lrs=phase1_hand_lrs, phase2_hand_lrs=phase2_hand_lrs, obj_lrs=obj_lrs,
                     obj_2half_lrs=obj_2half_lrs, noise_obj_lr1=noise_obj_lr1, noise_obj_lr2=noise_obj_lr2)
    state = dict(noise_pred_obj=noise_pred_obj, scale_hand=scale_hand, trans_hand=trans_hand, rotation_hand=rotation_hand,
                 scale_obj=scale_obj, trans_obj=trans_obj, rotation_obj=rotation_obj)
    groups: List[Dict] = []
    for name, table, key in _PHASES[phase]:
        leaf = state[name].clone().detach().requires_grad_(True)
        state[name] = leaf
        lr = lr_tables[table] if key is None else lr_tables[table][key]
        groups.append({"params": [leaf], "lr": lr})
    return (groups, state["noise_pred_obj"], state["scale_hand"], state["trans_hand"], state["rotation_hand"],
            state["scale_obj"], state["trans_obj"], state["rotation_obj"])
