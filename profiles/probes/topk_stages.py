"""Where k_topk_rows spends its time: the same 16384 x 10677 random score matrix through builds that stop after a
stage (-DHSK_TOPK_STOP=1: after the row's loads, 2: after the bound L, 3: after the candidates are collected).
usage: python profiles/probes/topk_stages.py lib1.so lib2.so ...   (run per library in a child process)"""
import os, subprocess, sys
if len(sys.argv) > 2:
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, lib], env=dict(os.environ, HSK_LIB_PATH=os.path.abspath(lib)))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hassaku_amd import hip_ops as ops
for rows, cols in ((16384, 10677), (16384, 16384)):
    x = torch.randn(rows, cols, device='cuda')
    for _ in range(3): ops.topk_dense(x, 100)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.topk_dense(x, 100)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e3
    e0.record()
    for _ in range(10): x.sum(1)
    e1.record(); torch.cuda.synchronize()
    t2 = e0.elapsed_time(e1) / 10 * 1e3
    print(os.path.basename(os.environ.get('HSK_LIB_PATH', 'default')), (rows, cols), 'topk %.1f us = %.2f TB/s;  torch row sum %.1f us' % (t, rows * cols * 4 / t / 1e6, t2))
