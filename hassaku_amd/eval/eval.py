"""Full-catalogue evaluation -- drop-in for eval/eval.py:14-118,211-258 of the reference.

`evaluate_recommender_algorithm(alg, eval_loader, evaluator, device, verbose)` keeps its signature.  For the
HIP matrix-factorisation model it does not densify anything on the host: per chunk of users it runs
hsk_mf_eval_topk (fp32-MFMA scores of the chunk against the item table, -inf on the user's excluded items
read from the exclude CSR, top-100) and hsk_rank_metrics (precision / recall / ndcg at 5,10,50,100 from
the ground-truth CSR), then accumulates per-group sums exactly as FullEvaluator does.  Any other
`RecommenderAlgorithm` goes through the generic dense path (predict -> mask -> eval_batch).
"""
from collections import defaultdict
from typing import Optional

import numpy as np
import torch

from hassaku_amd import hip_ops
from hassaku_amd.algorithms.base_classes import RecommenderAlgorithm, SGDBasedRecommenderAlgorithm
from hassaku_amd.eval.metrics import ndcg_at_k_batch, precision_at_k_batch, recall_at_k_batch
from hassaku_amd.utilities.utils import log_info_results

METRIC_NAMES = ('precision', 'recall', 'ndcg')


class FullEvaluator:
    """Accumulates metric sums for the 'all users' group (-1) and each user group; get_results() divides by
    the group sizes and resets (eval/eval.py:14-118)."""
    K_VALUES = [5, 10, 50, 100]

    def __init__(self, aggr_by_group: bool = True, n_groups: int = 0, user_to_user_group: Optional[torch.Tensor] = None):
        self.aggr_by_group = aggr_by_group
        self.n_groups = n_groups
        self.user_to_user_group = user_to_user_group
        self._reset_internal_dict()

    def _reset_internal_dict(self):
        self.group_metrics = defaultdict(lambda: defaultdict(float if self.aggr_by_group else list))
        self.n_entries = defaultdict(int)

    def get_n_groups(self):
        return self.n_groups

    def get_user_to_user_group(self):
        return self.user_to_user_group

    def _groups_of(self, u_idxs: torch.Tensor):
        """[(group id, boolean row selector or None for everyone)]"""
        sel = [(-1, None)]
        if self.get_n_groups() > 0:
            g = self.get_user_to_user_group().to(u_idxs.device)[u_idxs]
            sel += [(gi, g == gi) for gi in range(self.n_groups)]
        return sel

    def _accumulate(self, u_idxs: torch.Tensor, per_user: dict):
        """per_user: metric name -> tensor [batch].  Sums and group sizes stay on the device (fp64): the host reads
        them once, in get_results() -- not 12 x (1 + n_groups) times per batch."""
        for gi, rows in self._groups_of(u_idxs):
            n = u_idxs.shape[0] if rows is None else rows.sum()
            self.n_entries[gi] = self.n_entries[gi] + n
            for name, vals in per_user.items():
                v = vals if rows is None else vals[rows]
                if self.aggr_by_group:
                    self.group_metrics[gi][name] = self.group_metrics[gi][name] + v.double().sum()
                else:
                    self.group_metrics[gi][name] += [v.detach()]

    def eval_batch(self, u_idxs: torch.Tensor, logits: torch.Tensor, y_true: torch.Tensor):
        """Generic entry: dense logits [B, I] and labels [B, I] (eval/eval.py:54-99)."""
        ks = sorted(self.K_VALUES, reverse=True)
        if logits.is_cuda and logits.dtype == torch.float32:
            idx = hip_ops.topk_dense(logits.contiguous(), ks[0])[1]
        else:
            idx = torch.topk(logits, k=ks[0]).indices
        per_user = {}
        for k in ks:
            idx = idx[:, :k]
            for name, fn in zip(METRIC_NAMES, (precision_at_k_batch, recall_at_k_batch, ndcg_at_k_batch)):
                per_user[f'{name}@{k}'] = fn(logits, y_true, k, aggr_sum=False, idx_topk=idx).detach()
        self._accumulate(u_idxs, per_user)

    def eval_ranked(self, u_idxs: torch.Tensor, metrics: torch.Tensor, ks):
        """HIP entry: metrics [B, len(ks), 3] from hsk_rank_metrics."""
        per_user = {f'{name}@{k}': metrics[:, t, j] for t, k in enumerate(ks) for j, name in enumerate(METRIC_NAMES)}
        self._accumulate(u_idxs, per_user)

    def get_results(self):
        out = {}
        for gi, metrics in self.group_metrics.items():
            prefix = '' if gi == -1 else f'group_{gi}_'
            for name, acc in metrics.items():
                if self.aggr_by_group:
                    out[prefix + name] = float(acc) / float(self.n_entries[gi])
                else:
                    out[prefix + name] = torch.cat(acc).cpu().numpy()
        self._reset_internal_dict()
        return out


def _hip_mf_eval(alg, dataset, evaluator: FullEvaluator, device, chunk: int):
    arrays = dataset.device_arrays(device)
    user_emb, item_emb, ib, ub, gb = alg.tables()
    ks = sorted(evaluator.K_VALUES, reverse=True)
    k_max = ks[0]
    if dataset.n_items < k_max:
        raise ValueError(f'full evaluation needs at least {k_max} items (K_VALUES), got {dataset.n_items}')
    status = alg.status_word()
    for lo in range(0, dataset.n_users, chunk):
        u = torch.arange(lo, min(lo + chunk, dataset.n_users), device=device)
        # top-k selected inside the score GEMM: no [chunk, n_items] matrix exists (hsk_mf_eval_topk_fused)
        _, ids, _ = hip_ops.mf_eval_topk(user_emb, item_emb, ib, ub, gb, u, k_max, arrays['excl_indptr'],
                                         arrays['excl_indices'], status=status)
        met = hip_ops.rank_metrics(ids, u, arrays['label_indptr'], arrays['label_indices'], ks)
        evaluator.eval_ranked(u, met, ks)
    alg.check_indices()


def evaluate_recommender_algorithm(alg: RecommenderAlgorithm, eval_loader, evaluator: FullEvaluator, device='cpu',
                                   verbose=False):
    from hassaku_amd.algorithms.sgd_alg import SGDMatrixFactorization
    dataset = eval_loader.dataset
    if isinstance(alg, SGDMatrixFactorization):
        dev = next(alg.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('SGDMatrixFactorization evaluates on the HIP device only; move the model with '
                               '.to("cuda") (conf device: cuda)')
        with torch.no_grad():
            # wide catalogues select inside the GEMM (nothing of size chunk x n_items exists); narrow ones (fewer than
            # hip_ops.FUSED_TOPK_MIN_ITEMS columns) materialise chunk x n_items scores: at most 2 GB per chunk
            chunk = max(int(getattr(eval_loader, 'batch_size', 256) or 256), 16384)
            _hip_mf_eval(alg, dataset, evaluator, dev, chunk)
    elif isinstance(alg, SGDBasedRecommenderAlgorithm) and hasattr(alg, 'lookup'):
        # the other SGD models: item representations once, then every user batch against them (eval/eval.py:237-248);
        # top-k and metrics on the device as for any dense score matrix
        dev = next(alg.parameters()).device
        excl = dataset.exclude_csr
        with torch.no_grad():
            i_repr = alg.get_item_representations(torch.arange(dataset.n_items, device=dev))
            for u_idxs, _i_idxs, labels in eval_loader:
                u_dev = u_idxs.to(dev)
                out = alg.combine_user_item_representations(alg.get_user_representations(u_dev), i_repr).contiguous()
                rows = np.repeat(np.arange(len(u_idxs)), excl.row_lengths()[u_idxs.cpu().numpy()])
                cols = np.concatenate([excl.row(int(u)) for u in u_idxs]) if len(rows) else np.zeros(0, np.int64)
                out[torch.as_tensor(rows, device=dev), torch.as_tensor(cols, dtype=torch.int64, device=dev)] = -torch.inf
                evaluator.eval_batch(u_dev, out, labels.to(dev))
            alg.get_and_reset_other_loss()
            alg.check_indices()
    else:
        iterator = eval_loader
        if verbose:
            from tqdm import tqdm
            iterator = tqdm(eval_loader)
        excl = dataset.exclude_csr
        with torch.no_grad():
            for u_idxs, i_idxs, labels in iterator:
                out = alg.predict(u_idxs.to(device), i_idxs.to(device))
                if not isinstance(out, torch.Tensor):
                    out = torch.as_tensor(np.asarray(out))
                out = out.to(device)
                rows = np.repeat(np.arange(len(u_idxs)), excl.row_lengths()[u_idxs.cpu().numpy()])
                cols = np.concatenate([excl.row(int(u)) for u in u_idxs]) if len(rows) else np.zeros(0, np.int64)
                out[torch.as_tensor(rows, device=out.device), torch.as_tensor(cols, dtype=torch.int64, device=out.device)] = -torch.inf
                evaluator.eval_batch(u_idxs.to(device), out, labels.to(device))
    metrics_values = evaluator.get_results()
    log_info_results(metrics_values)
    return metrics_values
