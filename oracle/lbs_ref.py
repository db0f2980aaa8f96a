"""ORACLE (tests only): torch restatement of smplx==0.1.28 `lbs()` as used by MANOLayer(pose2rot=False)
(reference call site third_party/estimator/hamer/hamer/models/hamer.py:125-130, wrapper mano_wrapper.py:28-40;
SURVEY.md A.7).  PARITY UNPINNED: smplx is not vendored / installable here and the MANO assets are licence-gated,
so the algorithm is restated and exercised with a synthetic MANO-shaped model (followmyhold_amd.synthetic).
"""
import torch


def lbs(betas, rot_mats, model):
    """betas (B,10), rot_mats (B,16,3,3).  model: dict of tensors v_template (V,3), shapedirs (V,3,10),
    posedirs (135,3V), J_regressor (16,V), lbs_weights (V,16), parents (16).  Returns verts (B,V,3), joints (B,16,3)."""
    B = betas.shape[0]
    dt = betas.dtype
    vt, S, P = model["v_template"].to(dt), model["shapedirs"].to(dt), model["posedirs"].to(dt)
    Jr, W, parents = model["J_regressor"].to(dt), model["lbs_weights"].to(dt), [int(p) for p in model["parents"]]
    v_shaped = vt[None] + torch.einsum("bl,mkl->bmk", betas, S)
    J = torch.einsum("bik,ji->bjk", v_shaped, Jr)
    ident = torch.eye(3, dtype=dt)
    pose_feature = (rot_mats[:, 1:] - ident).reshape(B, -1)
    v_posed = v_shaped + (pose_feature @ P).reshape(B, -1, 3)
    # batch_rigid_transform
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    G = []
    for i in range(len(parents)):
        M = torch.zeros(B, 4, 4, dtype=dt)
        M[:, :3, :3] = rot_mats[:, i]
        M[:, :3, 3] = rel[:, i]
        M[:, 3, 3] = 1.0
        G.append(M if parents[i] < 0 else G[parents[i]] @ M)
    G = torch.stack(G, dim=1)
    posed_joints = G[:, :, :3, 3]
    Jh = torch.cat([J, torch.zeros(B, J.shape[1], 1, dtype=dt)], dim=2)[..., None]
    A = G - torch.nn.functional.pad(G @ Jh, [3, 0])
    T = (W[None] @ A.reshape(B, len(parents), 16)).reshape(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=dt)], dim=2)
    verts = (T @ vh[..., None])[:, :, :3, 0]
    return verts, posed_joints
