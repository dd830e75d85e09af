"""Batched ranking metrics with the reference's signatures (eval/metrics.py:4-105).

These are the generic entry points an evaluator for ANY algorithm can call with dense logits / labels.
The BPR-MF hot path does not go through them: it gets per-user metrics from hsk_rank_metrics, straight
from the top-k ids and the ground-truth CSR on the device.  Definitions (binary relevance):
  precision@k = hits/k;  recall@k = hits/|relevant| (0 when the user has no relevant item);
  ndcg@k = DCG/IDCG with discount 1/log2(rank+2), IDCG over min(k,|relevant|) ranks, clamped to <= 1.
"""
import torch


def _ranked_relevance(logits, y_true, k, idx_topk):
    if idx_topk is None:
        if logits.is_cuda and logits.dtype == torch.float32:
            from hassaku_amd import hip_ops
            idx_topk = hip_ops.topk_dense(logits.contiguous(), k)[1]
        else:
            idx_topk = torch.topk(logits, k=k).indices
    elif idx_topk.shape[-1] != k:
        raise AssertionError('Top-k indexes have different "k" compared to the parameter function')
    return torch.gather(y_true, -1, idx_topk.to(torch.int64))


def _discounts(k, device):
    return torch.log2(torch.arange(2, k + 2, device=device, dtype=torch.float32)).reciprocal()


def precision_at_k_batch(logits, y_true, k: int = 10, aggr_sum: bool = True, idx_topk=None):
    rel = _ranked_relevance(logits, y_true, k, idx_topk)
    out = rel.sum(-1) / k
    return out.sum() if aggr_sum else out


def recall_at_k_batch(logits, y_true, k: int = 10, aggr_sum: bool = True, idx_topk=None):
    rel = _ranked_relevance(logits, y_true, k, idx_topk)
    n_rel = y_true.sum(-1)
    out = torch.where(n_rel > 0, rel.sum(-1) / n_rel.clamp(min=1), torch.zeros_like(n_rel))
    return out.sum() if aggr_sum else out


def ndcg_at_k_batch(logits, y_true, k: int = 10, aggr_sum: bool = True, idx_topk=None):
    rel = _ranked_relevance(logits, y_true, k, idx_topk)
    disc = _discounts(k, rel.device)
    dcg = (rel * disc).sum(-1)
    ideal = torch.topk(y_true, k).values
    idcg = (ideal * disc).sum(-1)
    out = torch.where(idcg > 0, dcg / idcg.clamp(min=1e-30), torch.zeros_like(dcg)).clamp(max=1.)
    return out.sum() if aggr_sum else out
