"""Probe: where the fused selection overtakes the materialised path as the catalogue widens (U = 16384, D = 512).
   python profiles/probes/eval_width.py n_items [n_items ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from hassaku_amd import hip_ops as ops  # noqa: E402

dev = torch.device('cuda:0')
for n in sys.argv[1:]:
    bench.EVAL_SHAPES['w'] = (16384, int(n), 512, 100)
    ops.FUSED_TOPK_MIN_ITEMS = 1 << 30
    a = bench.run_eval('w', dev, chunk=16384)
    ops.FUSED_TOPK_MIN_ITEMS = 1024
    b = bench.run_eval('w', dev, chunk=16384)
    print(f'{n} items: materialised {a["users_per_s"] / 1e6:.3f} M users/s, fused {b["users_per_s"] / 1e6:.3f} M users/s', flush=True)
