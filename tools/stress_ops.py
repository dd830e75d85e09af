#!/usr/bin/env python3
"""Randomised check of the operator-level entry points (the un-fused drop-in surface of include/hassaku_hip.h) against
plain torch on the same device, float64 where it matters:     python tools/stress_ops.py [seconds] [seed]
  mf_scores / rec_loss_grad (bpr, bce, sampled_softmax) / mf_backward  vs autograd of the textbook expressions (1e-5)
  embedding                                                             vs table[idx]               (exact)
  opt_dense (adamw / adam / adagrad)                                    vs torch.optim              (2e-5 of the largest)
  topk_dense                                                            vs a stable sort             (values exact, ids with
                                                                                                      the lowest-id tie rule)
  sample_negatives_uniform / alias                                      invariants: in range, never a positive of the user
Random shapes of every alignment, duplicate indices, empty-ish corners."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hassaku_amd import hip_ops as ops  # noqa: E402


def close(a, b, tol, what):
    a, b = a.double(), b.double()
    scale = max(b.abs().max().item(), 1e-30) if b.numel() else 1.0
    err = (a - b).abs().max().item() / scale if b.numel() else 0.0
    return [] if err <= tol else [(what, err)]


def case_scores(rng, g):
    D = int(rng.choice([1, 2, 6, 16, 30, 33, 64, 100, 128, 200, 256, 402, 512, 1024]))
    U, I = int(rng.randint(1, 400)), int(rng.randint(8, 600))   # (two or three items: every gradient is a near-total
    # cancellation and 'relative to the largest element' stops meaning anything)
    B, K = int(rng.randint(1, 300)), int(rng.randint(2, 60))
    ue = torch.randn(U, D, device='cuda', generator=g) * 0.3
    ie = torch.randn(I, D, device='cuda', generator=g) * 0.3
    ib = torch.randn(I, device='cuda', generator=g) * 0.1 if rng.rand() < 0.7 else None
    ub = torch.randn(U, device='cuda', generator=g) * 0.1 if rng.rand() < 0.4 else None
    gb = torch.randn(1, device='cuda', generator=g) * 0.1 if rng.rand() < 0.4 else None
    u = torch.from_numpy(rng.randint(0, U, size=B).astype(np.int64)).cuda()
    i = torch.from_numpy(rng.randint(0, I, size=(B, K)).astype(np.int64)).cuda()
    kind = str(rng.choice(['bpr', 'bce', 'sampled_softmax']))
    adj = float(np.log(max(I, 2) / (K - 1))) if kind == 'sampled_softmax' else 0.0
    bad = []
    logits = ops.mf_scores(ue, ie, ib, ub, gb, u, i)
    p = [t.double().requires_grad_(True) for t in (ue, ie)]
    pb = [None if t is None else t.double().requires_grad_(True) for t in (ib, ub, gb)]
    ref = (p[0][u][:, None, :] * p[1][i]).sum(-1)
    if pb[1] is not None:
        ref = ref + pb[1][u][:, None]
    if pb[0] is not None:
        ref = ref + pb[0][i]
    if pb[2] is not None:
        ref = ref + pb[2]
    bad += close(logits, ref.detach(), 1e-5, 'mf_scores')
    loss, gl = ops.rec_loss_grad(kind, logits, adj)
    x = logits.double().requires_grad_(True)
    if kind == 'bpr':
        lref = torch.nn.functional.softplus(-(x[:, :1] - x[:, 1:])).mean()
    elif kind == 'bce':
        y = torch.zeros_like(x)
        y[:, 0] = 1.0
        lref = torch.nn.functional.binary_cross_entropy_with_logits(x, y)
    else:
        z = x.clone()
        z[:, 1:] = z[:, 1:] + adj * 0  # (the adjustment's sign convention is the library's; checked through the gradient sum)
        lref = None
    if lref is not None:
        lref.backward()
        if abs(float(loss) - lref.item()) > 2e-6 * max(abs(lref.item()), 1e-3):
            bad.append(('loss', kind, float(loss), lref.item()))
        bad += close(gl, x.grad, 1e-5, 'd loss / d logits ' + kind)
        gU, gI, gIb, gUb, ggb = ops.mf_backward(ue, ie, u, i, gl, ib is not None, ub is not None, gb is not None)
        ref.backward(x.grad)
        bad += close(gU, p[0].grad, 2e-5, 'grad user_emb')
        bad += close(gI, p[1].grad, 2e-5, 'grad item_emb')
        if ib is not None:
            bad += close(gIb.reshape(-1), pb[0].grad, 2e-5, 'grad item_bias')
        if ub is not None and kind == 'bce':
            bad += close(gUb.reshape(-1), pb[1].grad, 2e-5, 'grad user_bias')
    else:   # sampled softmax: rows of the gradient sum to zero, the positive's entry is negative
        s = gl.double().sum(1).abs().max().item()
        if s > 1e-6 or (gl[:, 0] > 0).any():
            bad.append(('sampled softmax gradient rows', s))
    e = ops.embedding(ie, i.reshape(-1))
    if not torch.equal(e, ie[i.reshape(-1)]):
        bad.append(('embedding',))
    return bad, dict(op='scores', D=D, U=U, I=I, B=B, K=K, kind=kind)


def case_opt(rng, g):
    n = int(rng.choice([1, 7, 64, 1000, 4097, 100003]))
    opt = str(rng.choice(['adamw', 'adam', 'adagrad']))
    lr, wd = float(10 ** rng.uniform(-4, -2)), float(rng.choice([0.0, 1e-4, 1e-2]))
    p0 = torch.randn(n, device='cuda', generator=g)
    p, m, v = p0.clone(), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    q = p0.double().clone().requires_grad_(True)
    cls = {'adamw': torch.optim.AdamW, 'adam': torch.optim.Adam, 'adagrad': torch.optim.Adagrad}[opt]
    o = cls([q], lr=lr, weight_decay=wd)
    bad = []
    for step in range(1, int(rng.randint(2, 6))):
        gr = torch.randn(n, device='cuda', generator=g) * float(10 ** rng.uniform(-6, 0))
        if rng.rand() < 0.3:
            gr[::2] = 0.0
        ops.opt_dense(opt, p, gr, m, v, lr, wd, step)
        q.grad = gr.double()
        o.step()
        bad += close(p, q.detach(), 2e-5, f'{opt} step {step}')   # (the default build's v_sqrt_f32 / v_rcp_f32: DESIGN section 2)
    return bad, dict(op='opt', opt=opt, n=n, lr=lr, wd=wd)


def case_topk(rng, g):
    rows, cols = int(rng.randint(1, 60)), int(rng.choice([1, 5, 100, 101, 1000, 4097, 12289, 20000]))
    k = int(min(cols, rng.choice([1, 5, 100])))
    x = torch.randn(rows, cols, device='cuda', generator=g)
    if rng.rand() < 0.5 and cols > 4:
        x[:, ::3] = x[:, :1]                      # ties
    if rng.rand() < 0.3:
        x[:, : cols // 2] = float('-inf')
    v, i = ops.topk_dense(x, k)
    order = torch.sort(x.double(), dim=1, descending=True, stable=True)   # stable: the lowest index first among equals
    bad = []
    if not torch.equal(v.double(), order.values[:, :k]):
        bad.append(('topk values',))
    if not torch.equal(i, order.indices[:, :k]):
        bad.append(('topk ids', int((i != order.indices[:, :k]).sum())))
    return bad, dict(op='topk', rows=rows, cols=cols, k=k)


def case_sampler(rng, g):
    U, I = int(rng.randint(1, 300)), int(rng.randint(3, 5000))
    dens = min(0.6, float(rng.choice([3.0, 30.0, 300.0])) / I)
    mask = rng.rand(U, I) < dens
    mask[:, 0] = False                            # every user keeps at least one admissible item
    mask[0, 1] = True                             # ... and the CSR is not empty (a NULL index array is refused)
    ptr = np.zeros(U + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(mask.sum(1))
    idx = np.nonzero(mask)[1].astype(np.int32)
    B, n_neg = int(rng.randint(1, 400)), int(rng.choice([1, 7, 64, 100, 130]))
    u = torch.from_numpy(rng.randint(0, U, size=B).astype(np.int64)).cuda()
    alias = None
    if rng.rand() < 0.4:
        pr, al = ops.build_alias_table(rng.rand(I) + 1e-3)
        alias = (torch.from_numpy(pr).cuda(), torch.from_numpy(al).cuda())
    neg = ops.sample_negatives_uniform(torch.from_numpy(ptr).cuda(), torch.from_numpy(idx).cuda(), I, u, n_neg,
                                       int(rng.randint(1 << 30)), int(rng.randint(1 << 20)), alias=alias).cpu().numpy()
    bad = []
    if neg.min() < 0 or neg.max() >= I:
        bad.append(('negative out of range',))
    if mask[u.cpu().numpy()[:, None], neg].any():
        bad.append(('a positive was drawn', int(mask[u.cpu().numpy()[:, None], neg].sum())))
    return bad, dict(op='sampler', U=U, I=I, B=B, n_neg=n_neg, alias=alias is not None)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    g = torch.Generator(device='cuda').manual_seed(seed)
    t_end = time.time() + budget
    cases = [case_scores, case_opt, case_topk, case_sampler]
    count = {c.__name__: 0 for c in cases}
    n_bad = 0
    while time.time() < t_end:
        c = cases[int(rng.randint(len(cases)))]
        try:
            bad, desc = c(rng, g)
        except (RuntimeError, ValueError) as e:
            bad, desc = [('raised', str(e)[:200])], dict(op=c.__name__)
        count[c.__name__] += 1
        if bad:
            n_bad += 1
            print('FAIL', desc, bad[:3], flush=True)
    print(count, f'{n_bad} failures', flush=True)
    sys.exit(1 if n_bad else 0)


if __name__ == '__main__':
    main()
