"""Probe: the fused (selection inside the GEMM) path forced onto the narrow ml10m-shaped catalogue.
   python profiles/probes/eval_fused_narrow.py chunk [chunk ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from hassaku_amd import hip_ops as ops  # noqa: E402

dev = torch.device('cuda:0')
for c in sys.argv[1:]:
    ops.FUSED_TOPK_MIN_ITEMS = 1 << 30
    a = bench.run_eval('ml10m', dev, chunk=int(c))
    ops.FUSED_TOPK_MIN_ITEMS = 1024
    b = bench.run_eval('ml10m', dev, chunk=int(c))
    print(f'ml10m chunk {c}: materialised {a["users_per_s"] / 1e6:.2f} M users/s, fused {b["users_per_s"] / 1e6:.2f} M users/s '
          f'(ndcg@10 {a["ndcg@10_check"]:.6f} / {b["ndcg@10_check"]:.6f})', flush=True)
