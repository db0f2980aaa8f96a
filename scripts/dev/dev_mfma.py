"""LBS pose-blend contraction on the f32 matrix cores (k_lbs_poseblend_mfma): run it at a few batch sizes so that a
rocprofv3 pass can report its duration and MFMA busy cycles (see profiles/README.md)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from followmyhold_amd import ops, synthetic
model = ops.LbsModel(synthetic.mano_like_model())
for B in (64, 1024, 8192):
    betas = torch.randn(B, 10, device="cuda")
    rot = torch.eye(3, device="cuda").repeat(B, 16, 1, 1).clone()
    for _ in range(20):
        ops.lbs(betas, rot, model, use_mfma=1)
    torch.cuda.synchronize()
print("done")
