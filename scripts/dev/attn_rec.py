"""bench.py's vae_attention record alone.  python scripts/dev/attn_rec.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
print(json.dumps(bench.vae_attention_record(torch, torch.device("cuda", 0))), flush=True)
