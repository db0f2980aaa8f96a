"""bench.py's lbs record alone.  python scripts/dev_lbs.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from followmyhold_amd import synthetic
r = bench.lbs_record(torch, np, synthetic, torch.device("cuda", 0))
for k, v in r.items():
    print(k, {a: round(b, 2) for a, b in v.items()} if isinstance(v, dict) else v)
