"""Probe: users/s of the lfm2b-shaped fused evaluation against the chunk of users handed to one call.
   python profiles/probes/eval_chunk.py [shape] chunk [chunk ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

args = sys.argv[1:]
shape = args.pop(0) if args and not args[0].isdigit() else 'lfm2b'
dev = torch.device('cuda:0')
for c in args:
    r = bench.run_eval(shape, dev, chunk=int(c))
    print(f'{shape} chunk {c}: {r["users_per_s"] / 1e3:.1f} k users/s', flush=True)
