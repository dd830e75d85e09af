"""Thin torch-tensor front end over the C ABI (include/hassaku_hip.h).

PyTorch is used here only as the owner of device memory and streams; every computation is a HIP
kernel of libhassaku_hip.so.  Shapes are validated on the host before any pointer is handed over.
"""
import ctypes
from typing import Optional, Sequence, Tuple

import torch

from hassaku_amd import _lib
from hassaku_amd._lib import HskBprmfState

ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-8  # torch.optim.AdamW defaults (train/trainer.py:52-53)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: Optional[torch.Tensor], dtype, name: str, shape: Optional[Tuple[int, ...]] = None, optional=False):
    if t is None:
        if optional:
            return
        raise ValueError(f'{name} must not be None')
    if not t.is_cuda:
        raise RuntimeError(f'{name} must live on the HIP device (got {t.device}); there is no CPU path')
    if t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f'{name} has shape {tuple(t.shape)}, expected {tuple(shape)}')


def new_status(device) -> torch.Tensor:
    return torch.zeros(1, dtype=torch.int32, device=device)


def raise_on_status(status: torch.Tensor, what: str = ''):
    """Synchronising check of the device status word."""
    s = int(status.item())
    if s & 1:
        raise IndexError(f'{what}: index out of range in self')
    if s & 2:
        raise RuntimeError(f'{what}: negative sampler gave up (a user interacted with ~every item)')


# ------------------------------------------------------------------------------------------------
# un-fused operators
# ------------------------------------------------------------------------------------------------
def mf_scores(user_emb, item_emb, item_bias, user_bias, global_bias, u_idx, i_idx, status=None):
    """logits[b,k] = <U[u_b], I[i_bk]> + biases.  u_idx [B] int64, i_idx [B,K] int64 -> [B,K] fp32."""
    _lib.require_gpu()
    lib = _lib.load()
    n_users, dim = user_emb.shape
    n_items = item_emb.shape[0]
    _chk(user_emb, torch.float32, 'user_emb')
    _chk(item_emb, torch.float32, 'item_emb', (n_items, dim))
    _chk(item_bias, torch.float32, 'item_bias', optional=True)
    _chk(user_bias, torch.float32, 'user_bias', optional=True)
    _chk(global_bias, torch.float32, 'global_bias', optional=True)
    if item_bias is not None and item_bias.numel() != n_items:
        raise ValueError('item_bias size mismatch')
    if user_bias is not None and user_bias.numel() != n_users:
        raise ValueError('user_bias size mismatch')
    _chk(u_idx, torch.int64, 'u_idx')
    _chk(i_idx, torch.int64, 'i_idx')
    if u_idx.dim() != 1 or i_idx.dim() != 2 or i_idx.shape[0] != u_idx.shape[0]:
        raise ValueError(f'u_idx {tuple(u_idx.shape)} / i_idx {tuple(i_idx.shape)}: expected [B] and [B,K]')
    B, K = i_idx.shape
    out = torch.empty((B, K), dtype=torch.float32, device=user_emb.device)
    # the kernel takes at most 65535 rows per launch
    for lo in range(0, B, 65535):
        hi = min(B, lo + 65535)
        _lib.check(lib.hsk_mf_scores(_p(user_emb), _p(item_emb), _p(item_bias), _p(user_bias), _p(global_bias),
                                     n_users, n_items, dim, _p(u_idx[lo:hi]), _p(i_idx[lo:hi]), hi - lo, K,
                                     _p(out[lo:hi]), _p(status), _stream()), 'hsk_mf_scores')
    return out


def bpr_loss_grad(logits: torch.Tensor, need_grad: bool = True):
    """-> (loss fp64 [1], grad_logits fp32 [B,K] or None)."""
    _lib.require_gpu()
    lib = _lib.load()
    _chk(logits, torch.float32, 'logits')
    if logits.dim() != 2 or logits.shape[1] < 2 or logits.shape[0] < 1:
        raise ValueError(f'logits must be [B>=1, K>=2], got {tuple(logits.shape)}')
    B, K = logits.shape
    loss = torch.empty(1, dtype=torch.float64, device=logits.device)
    ws = torch.empty(B, dtype=torch.float64, device=logits.device)
    grad = torch.empty_like(logits) if need_grad else None
    _lib.check(lib.hsk_bpr_loss_grad(_p(logits), B, K, _p(loss), _p(grad), _p(ws), _stream()), 'hsk_bpr_loss_grad')
    return loss, grad


LOSS_KINDS = {'bpr': 0, 'bce': 1, 'sampled_softmax': 2}   # HSK_LOSS_* of include/hassaku_hip.h


def rec_loss_grad(kind: str, logits: torch.Tensor, log_adjust: float = 0.0, need_grad: bool = True):
    """Any of the reference's three losses on logits [B, 1+N] -> (loss fp64 [1], grad_logits fp32 [B,K] or None)."""
    _lib.require_gpu()
    lib = _lib.load()
    _chk(logits, torch.float32, 'logits')
    if logits.dim() != 2 or logits.shape[1] < 2 or logits.shape[0] < 1:
        raise ValueError(f'logits must be [B>=1, K>=2], got {tuple(logits.shape)}')
    B, K = logits.shape
    loss = torch.empty(1, dtype=torch.float64, device=logits.device)
    ws = torch.empty(B, dtype=torch.float64, device=logits.device)
    grad = torch.empty_like(logits) if need_grad else None
    _lib.check(lib.hsk_rec_loss_grad(LOSS_KINDS[kind], _p(logits), B, K, float(log_adjust), _p(loss), _p(grad), _p(ws),
                                     _stream()), 'hsk_rec_loss_grad')
    return loss, grad


def mf_backward(user_emb, item_emb, u_idx, i_idx, grad_logits, want_item_bias, want_user_bias, want_global_bias,
                status=None):
    """Dense parameter gradients (what embedding_dense_backward would give)."""
    _lib.require_gpu()
    lib = _lib.load()
    n_users, dim = user_emb.shape
    n_items = item_emb.shape[0]
    _chk(user_emb, torch.float32, 'user_emb')
    _chk(item_emb, torch.float32, 'item_emb', (n_items, dim))
    _chk(u_idx, torch.int64, 'u_idx')
    _chk(i_idx, torch.int64, 'i_idx')
    B, K = i_idx.shape
    _chk(grad_logits, torch.float32, 'grad_logits', (B, K))
    if u_idx.shape != (B,):
        raise ValueError('u_idx / i_idx batch mismatch')
    if B > 65535:
        raise ValueError('mf_backward handles at most 65535 rows per call')
    dev = user_emb.device
    g_u = torch.empty_like(user_emb)
    g_i = torch.empty_like(item_emb)
    g_ib = torch.empty(n_items, dtype=torch.float32, device=dev) if want_item_bias else None
    g_ub = torch.empty(n_users, dtype=torch.float32, device=dev) if want_user_bias else None
    g_gb = torch.empty(1, dtype=torch.float32, device=dev) if want_global_bias else None
    _lib.check(lib.hsk_mf_backward(_p(user_emb), _p(item_emb), n_users, n_items, dim, _p(u_idx), _p(i_idx), B, K,
                                   _p(grad_logits), _p(g_u), _p(g_i), _p(g_ib), _p(g_ub), _p(g_gb), _p(status),
                                   _stream()), 'hsk_mf_backward')
    return g_u, g_i, g_ib, g_ub, g_gb


def bias_scores(item_bias, user_bias, global_bias, u_idx, i_idx, status=None):
    """Bias-only model (SGDBaseline, algorithms/sgd_alg.py:72-107): logits[b,k] = ub[u_b] + ib[i_bk] + gb."""
    _lib.require_gpu()
    lib = _lib.load()
    _chk(item_bias, torch.float32, 'item_bias')
    _chk(user_bias, torch.float32, 'user_bias')
    _chk(global_bias, torch.float32, 'global_bias', optional=True)
    _chk(u_idx, torch.int64, 'u_idx')
    _chk(i_idx, torch.int64, 'i_idx')
    if u_idx.dim() != 1 or i_idx.dim() != 2 or i_idx.shape[0] != u_idx.shape[0]:
        raise ValueError(f'u_idx {tuple(u_idx.shape)} / i_idx {tuple(i_idx.shape)}: expected [B] and [B,K]')
    B, K = i_idx.shape
    out = torch.empty((B, K), dtype=torch.float32, device=item_bias.device)
    for lo in range(0, B, 65535):
        hi = min(B, lo + 65535)
        _lib.check(lib.hsk_mf_scores(None, None, _p(item_bias), _p(user_bias), _p(global_bias), user_bias.numel(),
                                     item_bias.numel(), 0, _p(u_idx[lo:hi]), _p(i_idx[lo:hi]), hi - lo, K,
                                     _p(out[lo:hi]), _p(status), _stream()), 'hsk_mf_scores')
    return out


def bias_backward(n_users, n_items, u_idx, i_idx, grad_logits, want_global_bias=True, status=None):
    """Dense gradients of the bias-only model -> (g_item_bias [I], g_user_bias [U], g_global_bias [1] | None)."""
    _lib.require_gpu()
    lib = _lib.load()
    _chk(u_idx, torch.int64, 'u_idx')
    _chk(i_idx, torch.int64, 'i_idx')
    B, K = i_idx.shape
    _chk(grad_logits, torch.float32, 'grad_logits', (B, K))
    if B > 65535:
        raise ValueError('bias_backward handles at most 65535 rows per call')
    dev = grad_logits.device
    g_ib = torch.empty(n_items, dtype=torch.float32, device=dev)
    g_ub = torch.empty(n_users, dtype=torch.float32, device=dev)
    g_gb = torch.empty(1, dtype=torch.float32, device=dev) if want_global_bias else None
    _lib.check(lib.hsk_mf_backward(None, None, n_users, n_items, 0, _p(u_idx), _p(i_idx), B, K, _p(grad_logits), None,
                                   None, _p(g_ib), _p(g_ub), _p(g_gb), _p(status), _stream()), 'hsk_mf_backward')
    return g_ib, g_ub, g_gb


class _Embedding(torch.autograd.Function):
    """rows = table[idx] through hsk_embedding_gather; backward = hsk_embedding_backward (dense gradient, duplicates
    summed in ascending position -- nn.Embedding(sparse=False) semantics, deterministic)."""

    @staticmethod
    def forward(ctx, table, idx, status):
        _lib.require_gpu()
        lib = _lib.load()
        _chk(table, torch.float32, 'embedding table')
        _chk(idx, torch.int64, 'embedding indices')
        n_rows, dim = table.shape
        flat = idx.reshape(-1)
        out = torch.empty((flat.numel(), dim), dtype=torch.float32, device=table.device)
        _lib.check(lib.hsk_embedding_gather(_p(table), n_rows, dim, _p(flat), flat.numel(), _p(out), _p(status),
                                            _stream()), 'hsk_embedding_gather')
        ctx.save_for_backward(flat)
        ctx.shape, ctx.status = (n_rows, dim), status
        return out.view(tuple(idx.shape) + (dim,))

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        (flat,) = ctx.saved_tensors
        n_rows, dim = ctx.shape
        g = grad_out.reshape(-1, dim).contiguous()
        grad_table = torch.empty((n_rows, dim), dtype=torch.float32, device=g.device)
        if flat.numel() == 0:
            return grad_table.zero_(), None, None
        nbytes = lib.hsk_embedding_backward_ws_bytes(n_rows, flat.numel())
        if nbytes <= 0:
            raise ValueError('embedding table too large for the deterministic index sort')
        ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
        _lib.check(lib.hsk_embedding_backward(_p(g), _p(flat), flat.numel(), n_rows, dim, _p(grad_table), _p(ws), nbytes,
                                              _p(ctx.status), _stream()), 'hsk_embedding_backward')
        return grad_table, None, None


def embedding(table: torch.Tensor, idx: torch.Tensor, status=None) -> torch.Tensor:
    """table[idx] -> idx.shape + (dim,), differentiable wrt `table` (dense gradient)."""
    return _Embedding.apply(table, idx.contiguous(), status)


def adamw_dense(p, g, m, v, lr, wd, step, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=ADAM_EPS):
    """In-place dense AdamW step (step is 1-based)."""
    _lib.require_gpu()
    lib = _lib.load()
    _chk(p, torch.float32, 'p')
    _chk(m, torch.float32, 'm', tuple(p.shape))
    _chk(v, torch.float32, 'v', tuple(p.shape))
    _chk(g, torch.float32, 'g', tuple(p.shape), optional=True)
    _lib.check(lib.hsk_adamw_dense(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, wd, step, _stream()),
               'hsk_adamw_dense')


LAZY_USERS_MIN_ELEMENTS = 4_000_000   # user tables above this many elements are updated lazily (BprMfFusedState)
OPT_KINDS = {'adamw': 0, 'adam': 1, 'adagrad': 2}            # HSK_OPT_* (conf['optimizer'], train/trainer.py:48-53)
ADAGRAD_EPS = 1e-10                                           # torch.optim.Adagrad default


def opt_eps(optimizer: str, eps=None) -> float:
    """torch's default eps of the optimiser unless given."""
    if eps is not None:
        return eps
    return ADAGRAD_EPS if optimizer == 'adagrad' else ADAM_EPS


def opt_dense(optimizer, p, g, m, v, lr, wd, step, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=None):
    """In-place dense step of 'adamw' | 'adam' | 'adagrad' (torch.optim defaults; weight_decay = L2 for the last
    two).  v is exp_avg_sq / adagrad's state_sum; step is 1-based."""
    _lib.require_gpu()
    lib = _lib.load()
    if optimizer not in OPT_KINDS:
        raise ValueError(f'Optimizer {optimizer} not yet implemented')
    _chk(p, torch.float32, 'p')
    _chk(m, torch.float32, 'm', tuple(p.shape))
    _chk(v, torch.float32, 'v', tuple(p.shape))
    _chk(g, torch.float32, 'g', tuple(p.shape), optional=True)
    _lib.check(lib.hsk_opt_dense(OPT_KINDS[optimizer], _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2,
                                 opt_eps(optimizer, eps), wd, step, _stream()), 'hsk_opt_dense')


def build_alias_table(p):
    """Walker/Vose alias table of a discrete distribution p (host, float64) -> (prob float32 [n], alias int32 [n]):
    draw a uniform column j, keep j with probability prob[j], else take alias[j]."""
    import numpy as np
    p = np.asarray(p, dtype=np.float64)
    n = len(p)
    if n == 0 or p.min() < 0 or not np.isfinite(p).all() or p.sum() <= 0:
        raise ValueError('alias table needs a non-empty, non-negative, finite distribution')
    scaled = p / p.sum() * n
    prob, alias = np.ones(n, dtype=np.float64), np.arange(n, dtype=np.int32)
    small = [i for i in range(n) if scaled[i] < 1.0]
    large = [i for i in range(n) if scaled[i] >= 1.0]
    while small and large:
        s, g = small.pop(), large.pop()
        prob[s], alias[s] = scaled[s], g
        scaled[g] -= 1.0 - scaled[s]
        (small if scaled[g] < 1.0 else large).append(g)
    return prob.astype(np.float32), alias


def sample_negatives_uniform(csr_indptr, csr_indices, n_items, u_idx, n_neg, seed, stream_id=0, status=None,
                             alias=None):
    """neg[b,n] ~ U{[0,n_items) minus the CSR row of u_b}  (alias=None), or the item distribution of the alias table
    (alias = (prob f32 [I], idx i32 [I]) device tensors) restricted the same way.  -> int64 [B, n_neg]."""
    _lib.require_gpu()
    lib = _lib.load()
    _chk(csr_indptr, torch.int64, 'csr_indptr')
    _chk(csr_indices, torch.int32, 'csr_indices')
    _chk(u_idx, torch.int64, 'u_idx')
    n_users = csr_indptr.numel() - 1
    B = u_idx.numel()
    out = torch.empty((B, n_neg), dtype=torch.int64, device=u_idx.device)
    if alias is None:
        _lib.check(lib.hsk_sample_negatives_uniform(_p(csr_indptr), _p(csr_indices), n_users, n_items, _p(u_idx), B,
                                                    n_neg, seed, stream_id, _p(out), _p(status), _stream()),
                   'hsk_sample_negatives_uniform')
    else:
        prob, idx = alias
        _chk(prob, torch.float32, 'alias prob', (n_items,))
        _chk(idx, torch.int32, 'alias idx', (n_items,))
        _lib.check(lib.hsk_sample_negatives_alias(_p(csr_indptr), _p(csr_indices), n_users, n_items, _p(prob), _p(idx),
                                                  _p(u_idx), B, n_neg, seed, stream_id, _p(out), _p(status),
                                                  _stream()), 'hsk_sample_negatives_alias')
    return out


# ------------------------------------------------------------------------------------------------
# fused training step
# ------------------------------------------------------------------------------------------------
class BprMfFusedState:
    """Owns the torch tensors the C `hsk_bprmf_state` points at (moments, workspace, loss, status).

    Parameters are borrowed from the model (`nn.Parameter.data`): the fused step writes into the model's own
    tensors.  With the lazy AdamW (lazy_users, the default; lazy_items on large catalogues) a row outside the batch
    keeps up to 63 pending zero-gradient steps until it is next touched: call flush() before reading the tables
    (state_dict(), save_model_to_path(), evaluation) -- Trainer does.  One instance per (model, hyper-parameters).
    """

    def __init__(self, user_emb, item_emb, item_bias=None, user_bias=None, global_bias=None, *, lr, wd,
                 max_batch, max_cols, beta1=ADAM_BETA1, beta2=ADAM_BETA2, eps=None, seed=0,
                 csr_indptr=None, csr_indices=None, coo_user=None, coo_item=None, lazy_users='auto', overlap=True,
                 loss='bpr', log_adjust=0.0, alias=None, optimizer='adamw', lazy_items='auto', graph_chunk=0,
                 flush_every=0):
        _lib.require_gpu()
        if optimizer not in OPT_KINDS:
            raise ValueError(f'Optimizer {optimizer} not yet implemented')
        eps = opt_eps(optimizer, eps)
        self.lib = _lib.load()
        n_users, dim = user_emb.shape
        n_items = item_emb.shape[0]
        _chk(user_emb, torch.float32, 'user_emb')
        _chk(item_emb, torch.float32, 'item_emb', (n_items, dim))
        for t, name, n in ((item_bias, 'item_bias', n_items), (user_bias, 'user_bias', n_users),
                           (global_bias, 'global_bias', 1)):
            _chk(t, torch.float32, name, optional=True)
            if t is not None and t.numel() != n:
                raise ValueError(f'{name} has {t.numel()} elements, expected {n}')
        dev = user_emb.device
        self.device = dev
        self.params = dict(user_emb=user_emb, item_emb=item_emb, item_bias=item_bias, user_bias=user_bias,
                           global_bias=global_bias)
        self.m = {k: (torch.zeros_like(t) if t is not None else None) for k, t in self.params.items()}
        self.v = {k: (torch.zeros_like(t) if t is not None else None) for k, t in self.params.items()}
        self.loss_out = torch.zeros(2, dtype=torch.float64, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.max_batch, self.max_cols = int(max_batch), int(max_cols)
        nbytes = self.lib.hsk_bprmf_workspace_bytes(n_users, n_items, dim, self.max_batch, self.max_cols)
        if nbytes <= 0:
            raise ValueError('invalid workspace request')
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.csr_indptr, self.csr_indices, self.coo_user, self.coo_item = csr_indptr, csr_indices, coo_user, coo_item
        if csr_indptr is not None:
            _chk(csr_indptr, torch.int64, 'csr_indptr', (n_users + 1,))
            _chk(csr_indices, torch.int32, 'csr_indices')
            _chk(coo_user, torch.int32, 'coo_user')
            _chk(coo_item, torch.int32, 'coo_item', tuple(coo_user.shape))

        st = HskBprmfState()
        for k, t in self.params.items():
            setattr(st, k, _p(t))
            setattr(st, 'm_' + k, _p(self.m[k]))
            setattr(st, 'v_' + k, _p(self.v[k]))
        st.n_users, st.n_items, st.dim = n_users, n_items, dim
        st.lr, st.beta1, st.beta2, st.eps, st.wd = lr, beta1, beta2, eps, wd
        st.opt_kind = OPT_KINDS[optimizer]
        if lazy_items == 'auto':   # worth it when most item rows are outside every batch AND the dense sweep is not
            # small change anyway (a table of a few MB is swept in microseconds, inside the item pass's launch)
            lazy_items = (dim % 2 == 0 and n_items >= 2 * self.max_batch * self.max_cols
                          and n_items * dim > LAZY_USERS_MIN_ELEMENTS)
        st.lazy_items = 1 if lazy_items else 0
        st.step = 0
        st.csr_indptr, st.csr_indices = _p(csr_indptr), _p(csr_indices)
        st.coo_user, st.coo_item = _p(coo_user), _p(coo_item)
        st.nnz = 0 if coo_user is None else coo_user.numel()
        st.seed = seed & 0xFFFFFFFFFFFFFFFF
        st.workspace, st.workspace_bytes = _p(self.workspace), nbytes
        st.max_batch, st.max_cols = self.max_batch, self.max_cols
        # exact lazy AdamW on user rows (bit-identical to the dense sweep after flush()); dense when False.  'auto':
        # a table of only a few batches' worth of rows is swept densely inside the item pass's launch (measured at the
        # ml100k shape, U = 943, B = 128: 17.2 vs 19.4 us per step); beyond that only the batch's rows are touched
        # (ml1m shape, U = 6040: 31.6 vs 35.9 us)
        if lazy_users == 'auto':
            lazy_users = n_users > 8 * self.max_batch
        st.lazy_users = 1 if lazy_users else 0
        # steps_sampled() replays its steady-state loop as HIP graphs of this many steps (0: default 64, < 0: never)
        st.graph_chunk = int(graph_chunk)
        st.catchup_apart = 0
        # steps between the periodic sweeps of the lazily updated tables; 0: from the table / batch sizes
        # (hsk_bprmf_flush_cadence; a speed matter only -- every cadence gives the same bits)
        st.flush_every, st.ws_sharded = int(flush_every), 0
        st.timing_mask = 0
        if loss not in LOSS_KINDS:
            raise ValueError(f'unknown loss {loss!r}')
        if loss == 'bce' and (user_bias is not None or global_bias is not None):
            raise ValueError('the fused bce step treats user/global bias as gradient-free; use the autograd path')
        st.loss_kind, st.ssm_log_adjust = LOSS_KINDS[loss], float(log_adjust)
        self.alias = alias   # (prob f32 [I], idx i32 [I]) for train_neg_strategy 'popular', None = uniform
        if alias is not None:
            _chk(alias[0], torch.float32, 'alias prob', (n_items,))
            _chk(alias[1], torch.int32, 'alias idx', (n_items,))
        st.alias_prob, st.alias_idx = (None, None) if alias is None else (_p(alias[0]), _p(alias[1]))
        st.timing = None
        st.timing_every = 1
        self._timing = None
        # optional side stream: the batch named by hint_next() is sampled and item-sorted there while the current
        # step's item / user passes run (cross-step prefetch; results identical to the un-hinted sequence)
        self._aux = self.lib.hsk_aux_create() if overlap else None
        self._hint_order = None
        st.aux = self._aux
        st.loss_out, st.status = _p(self.loss_out), _p(self.status)
        self.st = st
        _lib.check(self.lib.hsk_bprmf_init_workspace(ctypes.byref(st), _stream()), 'hsk_bprmf_init_workspace')

    @property
    def step_count(self) -> int:
        return int(self.st.step)

    def step(self, u_idx: torch.Tensor, i_idx: torch.Tensor):
        """One fused step on a loader-provided batch (u_idx [B] int64, i_idx [B,1+N] int64)."""
        _chk(u_idx, torch.int64, 'u_idx')
        _chk(i_idx, torch.int64, 'i_idx')
        if i_idx.dim() != 2 or u_idx.shape != (i_idx.shape[0],):
            raise ValueError(f'u_idx {tuple(u_idx.shape)} / i_idx {tuple(i_idx.shape)}: expected [B] and [B,1+N]')
        B, K = i_idx.shape
        _lib.check(self.lib.hsk_bprmf_train_step(ctypes.byref(self.st), _p(u_idx), _p(i_idx), B, K, _stream()),
                   'hsk_bprmf_train_step')

    def step_sampled(self, order: Optional[torch.Tensor], start: int, batch: int, n_neg: int):
        """One fused step; positives = interactions order[start:start+batch], negatives drawn on device."""
        if order is not None:
            _chk(order, torch.int64, 'order')
            if start + batch > order.numel():
                raise ValueError('order too short')
        _lib.check(self.lib.hsk_bprmf_train_step_sampled(ctypes.byref(self.st), _p(order), start, batch, n_neg,
                                                         _stream()), 'hsk_bprmf_train_step_sampled')

    def steps_sampled(self, order: Optional[torch.Tensor], start: int, n_steps: int, batch: int, n_neg: int):
        """n_steps fused steps on consecutive batches of `batch` positives starting at order[start] -- the epoch's
        inner loop issued from C (each step hints the next one to the prefetch)."""
        if order is not None:
            _chk(order, torch.int64, 'order')
            if start + n_steps * batch > order.numel():
                raise ValueError('order too short')
        _lib.check(self.lib.hsk_bprmf_train_steps(ctypes.byref(self.st), _p(order), start, n_steps, batch, n_neg,
                                                  _stream()), 'hsk_bprmf_train_steps')

    def flush_cadence(self, batch: Optional[int] = None):
        """(steps between sweeps of the user table, of the item table) for batches of `batch` positives;
        2**30 = never (only flush() sweeps)."""
        import math
        b = self.max_batch if batch is None else int(batch)
        n_items = self.params['item_emb'].shape[0]
        touched = max(1, int(n_items * (1.0 - math.exp(-b * self.max_cols / n_items))))
        ref = ctypes.byref(self.st)
        return int(self.lib.hsk_bprmf_flush_cadence(ref, 0, b)), int(self.lib.hsk_bprmf_flush_cadence(ref, 1, touched))

    def graph_replays(self) -> int:
        """Runs of steps_sampled() issued as replayed HIP graphs so far."""
        return int(self.lib.hsk_bprmf_graph_replays(ctypes.byref(self.st)))

    def pipelined_steps(self) -> int:
        """Steps of steps_sampled() issued with the next batches' preparation riding in the steps' own launches."""
        return int(self.lib.hsk_bprmf_pipelined_steps(ctypes.byref(self.st)))

    def hint_next(self, order: Optional[torch.Tensor], start: int, batch: int, n_neg: int):
        """Name the batch of the NEXT step_sampled call so that the step issued now prepares it on the side stream.
        No-op without overlap=True.  batch <= 0 clears a pending hint."""
        if self._aux is None:
            return
        if order is not None and batch > 0:
            _chk(order, torch.int64, 'order')
            if start + batch > order.numel():
                raise ValueError('order too short')
        self._hint_order = order                      # keep the tensor alive until the hinted step was issued
        _lib.check(self.lib.hsk_bprmf_hint_next(ctypes.byref(self.st), _p(order), start, batch, n_neg),
                   'hsk_bprmf_hint_next')

    def hint_after_run(self, order: Optional[torch.Tensor], start: int, batch: int, n_neg: int, n_batches: int = 1):
        """Name the `n_batches` consecutive batches that follow the NEXT steps_sampled() run: the run's last steps prepare
        them (side stream, or -- large batches -- riding in the steps' own launches, which works two batches ahead), so
        an epoch issued in several runs keeps its preparation pipeline full.  No-op without overlap=True."""
        if self._aux is None:
            return
        self._hint_order = order
        _lib.check(self.lib.hsk_bprmf_hint_after_run_n(ctypes.byref(self.st), _p(order), start, batch, n_neg,
                                                       int(n_batches)), 'hsk_bprmf_hint_after_run_n')

    def last_batch(self, batch: int, n_cols: int):
        u = torch.empty(batch, dtype=torch.int64, device=self.device)
        i = torch.empty((batch, n_cols), dtype=torch.int64, device=self.device)
        _lib.check(self.lib.hsk_bprmf_last_batch(ctypes.byref(self.st), batch, n_cols, _p(u), _p(i), _stream()),
                   'hsk_bprmf_last_batch')
        return u, i

    def batch_columns(self, batch: int, n_cols: int) -> int:
        """Columns of the step's internal batch rows (n_cols, or n_cols + P - 1 under the item-partitioned forward)."""
        return int(self.lib.hsk_bprmf_batch_columns(ctypes.byref(self.st), batch, n_cols))

    def last_sort(self, n_entries: int):
        """(perm int32 [n_entries], offsets int32 [n_items + 1]): the item-major index of the last step's batch."""
        perm = torch.empty(n_entries, dtype=torch.int32, device=self.device)
        offs = torch.empty(self.params['item_emb'].shape[0] + 1, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.hsk_bprmf_last_sort(ctypes.byref(self.st), n_entries, _p(perm), _p(offs), _stream()),
                   'hsk_bprmf_last_sort')
        return perm, offs

    STAGES = ('prep', 'scan', 'scatter', 'fwd', 'item', 'user', 'finish')

    def enable_timing(self, stages=('fwd',), every=1):
        """Bracket the named stages of every `every`-th following step with HIP events on the launch stream."""
        if not getattr(self, '_timing', None):
            self._timing = self.lib.hsk_timing_create()
        mask = 0
        for s in stages:
            mask |= 1 << self.STAGES.index(s)
        self.st.timing = self._timing
        self.st.timing_mask = mask
        self.st.timing_every = int(every)

    def disable_timing(self):
        self.st.timing_mask = 0

    def collect_timing(self):
        """-> {stage: (total_ms, n_samples)}; waits for the recorded events."""
        if not getattr(self, '_timing', None):
            return {}
        n = len(self.STAGES)
        ms = (ctypes.c_double * n)()
        cnt = (ctypes.c_int64 * n)()
        _lib.check(self.lib.hsk_timing_collect(self._timing, ms, cnt), 'hsk_timing_collect')
        return {s: (ms[i], cnt[i]) for i, s in enumerate(self.STAGES) if cnt[i] > 0}

    def __del__(self):
        t = getattr(self, '_timing', None)
        if t:
            try:
                self.lib.hsk_timing_destroy(t)
            except Exception:
                pass
            self._timing = None
        a = getattr(self, '_aux', None)
        if a:
            try:
                torch.cuda.synchronize()
                self.lib.hsk_aux_destroy(a)
            except Exception:
                pass
            self._aux = None

    def flush(self):
        _lib.check(self.lib.hsk_bprmf_flush(ctypes.byref(self.st), _stream()), 'hsk_bprmf_flush')

    def set_hyper(self, lr=None, wd=None, beta1=None, beta2=None, eps=None):
        """Change the optimiser's hyper-parameters between steps (an LR schedule).  The library freezes them when the
        workspace is prepared -- the per-step Adam scalars of the lazy replay and of replayed graphs are a device table
        -- and refuses a step that finds them changed; this brings every row up to date under the OLD values (dense
        semantics: a pending zero-gradient step belongs to the schedule that was in force when it was due), installs the
        new ones and prepares the workspace again.  Loss sums and the step counter carry over."""
        self.flush()
        self.check_status()
        loss_sum = self.loss_out.clone()
        for name, val in (('lr', lr), ('wd', wd), ('beta1', beta1), ('beta2', beta2), ('eps', eps)):
            if val is not None:
                setattr(self.st, name, float(val))
        _lib.check(self.lib.hsk_bprmf_init_workspace(ctypes.byref(self.st), _stream()), 'hsk_bprmf_init_workspace')
        self.loss_out.copy_(loss_sum)

    def last_loss(self) -> float:
        return float(self.loss_out[0].item())

    def pop_loss_sum(self) -> float:
        """Sum of the per-step losses since the last call (one host sync)."""
        s = float(self.loss_out[1].item())
        self.loss_out[1].zero_()
        return s

    def check_status(self, what='fused BPR-MF step'):
        raise_on_status(self.status, what)


# ------------------------------------------------------------------------------------------------
# evaluation
# ------------------------------------------------------------------------------------------------
FUSED_TOPK_MAX_K = 128     # hsk_mf_eval_topk_fused (HSK_SEL_KMAX)
FUSED_TOPK_MIN_ITEMS = 20480   # below this many columns the materialised path is the faster one.  Round 4, with the in-GEMM
                               # selection on the 256 x 256 core, seeded and shared thresholds (csrc/hsk_eval_fused.hip:
                               # k_score_topk_wide): materialised / fused M users/s at U = 16 384 -- 16 384 items 6.85 / 6.15,
                               # 24 576: 3.88 / 4.38, 32 768: 2.98 / 3.57 (round 3's 128 x 128 selection crossed at 36 864)
_fused_ws = {}             # device -> scratch of the fused selection, grown on demand
_planes_ws = {}            # device -> scratch of the materialised path's operand pieces (fp16 pairs or bf16 triples), grown on demand


EVAL_ARITH_FP32, EVAL_ARITH_BF16X3, EVAL_ARITH_F16X2 = 0, 1, 2
EVAL_ARITH_DEFAULT = EVAL_ARITH_F16X2


def set_eval_arith(form=EVAL_ARITH_DEFAULT):
    """Arithmetic of the score GEMMs (materialised and fused top-k): 2 (default) two fp16 pieces per fp32 operand and three
    products on the 256 x 256 kernels (calls without the pieces' scratch or with rows that are not 16-byte aligned run
    form 1), 1 three bf16 pieces and six products, 0 exact fp32 MFMA; see hsk_eval_set_arith in include/hassaku_hip.h."""
    _lib.load().hsk_eval_set_arith(int(form))


def mf_eval_topk(user_emb, item_emb, item_bias, user_bias, global_bias, u_idx, k, excl_indptr=None, excl_indices=None,
                 item_begin=0, item_count=None, scores_ws=None, status=None, item_shard=False, n_items_global=None,
                 want_scores=None, presplit=True):
    """Top-k of one item shard's masked scores.  -> (vals [R,k] f32, idx [R,k] i32 global, scores or None).
    presplit=False withholds the scratch for the operands' pieces: the GEMM then cuts every tile into three bf16 pieces in
    its loop (form 1's bits whatever form is set; the form the library falls back to by itself, kept reachable for the
    parity test).
    want_scores=False: the selection runs inside the score GEMM and the score matrix is never formed (scores = None).
    want_scores=True (or a `scores_ws` buffer, k == 0, k > 128): the [R, item_count] matrix is materialised and
    returned.  None (default): whichever is faster for the shard width (FUSED_TOPK_MIN_ITEMS); same results.
    item_shard=True: `item_emb` / `item_bias` are the PHYSICAL shard (rows item_begin .. item_begin + item_count of a
    catalogue of n_items_global items); the library is handed the virtual base of the catalogue and only ever
    dereferences the shard's rows (include/hassaku_hip.h)."""
    _lib.require_gpu()
    lib = _lib.load()
    n_users, dim = user_emb.shape
    n_items = item_emb.shape[0]
    if item_shard:
        if item_count is None:
            item_count = n_items
        if item_count != n_items or n_items_global is None or item_begin + item_count > n_items_global:
            raise ValueError('item_shard: item_emb must hold exactly rows [item_begin, item_begin + item_count)')
    elif item_count is None:
        item_count = n_items - item_begin
    _chk(user_emb, torch.float32, 'user_emb')
    _chk(item_emb, torch.float32, 'item_emb', (n_items, dim))
    _chk(item_bias, torch.float32, 'item_bias', optional=True)
    if item_bias is not None and item_bias.numel() != n_items:
        raise ValueError('item_bias size mismatch')
    _chk(user_bias, torch.float32, 'user_bias', optional=True)
    _chk(global_bias, torch.float32, 'global_bias', optional=True)
    _chk(u_idx, torch.int64, 'u_idx')
    if excl_indptr is not None:
        _chk(excl_indptr, torch.int64, 'excl_indptr', (n_users + 1,))
        _chk(excl_indices, torch.int32, 'excl_indices')
    R = u_idx.numel()
    dev = user_emb.device
    if want_scores is None:
        want_scores = scores_ws is not None or item_count < FUSED_TOPK_MIN_ITEMS
    if not want_scores and scores_ws is None and 1 <= k <= FUSED_TOPK_MAX_K and R > 0:
        if presplit:
            need = lib.hsk_mf_eval_fused_ws_bytes_dim(R, item_count, k, dim)   # with room for the operands' pieces
        else:
            need = lib.hsk_mf_eval_fused_ws_bytes(R, item_count, k)
        if need <= 0:
            raise ValueError('invalid fused top-k request')
        ws = _fused_ws.get(dev)
        if ws is None or ws.numel() < need:
            ws = _fused_ws[dev] = torch.empty(need, dtype=torch.uint8, device=dev)
        vals = torch.empty((R, k), dtype=torch.float32, device=dev)
        idx = torch.empty((R, k), dtype=torch.int32, device=dev)
        p_emb, p_bias, n_all = _p(item_emb), _p(item_bias), n_items
        if item_shard:
            p_emb -= 4 * dim * item_begin
            p_bias = None if p_bias is None else p_bias - 4 * item_begin
            n_all = n_items_global
        _lib.check(lib.hsk_mf_eval_topk_fused(_p(user_emb), p_emb, p_bias, _p(user_bias), _p(global_bias), n_users,
                                              n_all, dim, _p(u_idx), R, item_begin, item_count, _p(excl_indptr),
                                              _p(excl_indices), k, _p(ws), ws.numel() if presplit else need, _p(vals),
                                              _p(idx), _p(status),
                                              _stream()), 'hsk_mf_eval_topk_fused')
        return vals, idx, None
    if scores_ws is None:
        scores_ws = torch.empty((R, item_count), dtype=torch.float32, device=dev)
    else:
        _chk(scores_ws, torch.float32, 'scores_ws')
        if scores_ws.numel() < R * item_count:
            raise ValueError('scores_ws too small')
    vals = torch.empty((R, k), dtype=torch.float32, device=dev)
    idx = torch.empty((R, k), dtype=torch.int32, device=dev)
    p_emb, p_bias = _p(item_emb), _p(item_bias)
    if item_shard:
        p_emb -= 4 * dim * item_begin
        p_bias = None if p_bias is None else p_bias - 4 * item_begin
        n_items = n_items_global
    # scratch for the operands' pieces (split once per call instead of once per tile in the GEMM loop)
    need = lib.hsk_mf_eval_planes_bytes(R, item_count, dim)
    planes = _planes_ws.get(dev)
    if planes is None or planes.numel() < need:
        planes = _planes_ws[dev] = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.check(lib.hsk_mf_eval_topk_planes(_p(user_emb), p_emb, p_bias, _p(user_bias), _p(global_bias),
                                           n_users, n_items, dim, _p(u_idx), R, item_begin, item_count,
                                           _p(excl_indptr), _p(excl_indices), k, _p(scores_ws),
                                           _p(planes) if presplit else None, planes.numel() if presplit else 0,
                                           _p(vals), _p(idx), _p(status), _stream()),
               'hsk_mf_eval_topk_planes')
    return vals, idx, scores_ws


def topk_dense(logits: torch.Tensor, k: int):
    """(values, indices int64) of the k largest entries per row; ties broken by lower index."""
    _lib.require_gpu()
    lib = _lib.load()
    _chk(logits, torch.float32, 'logits')
    if logits.dim() != 2:
        raise ValueError('logits must be 2-D')
    rows, cols = logits.shape
    vals = torch.empty((rows, k), dtype=torch.float32, device=logits.device)
    idx = torch.empty((rows, k), dtype=torch.int64, device=logits.device)
    _lib.check(lib.hsk_topk_dense(_p(logits), rows, cols, cols, k, _p(vals), _p(idx), _stream()), 'hsk_topk_dense')
    return vals, idx


def topk_merge(vals: torch.Tensor, idx: torch.Tensor):
    """[P,R,k] candidate lists -> global top-k [R,k]."""
    _lib.require_gpu()
    lib = _lib.load()
    _chk(vals, torch.float32, 'vals')
    _chk(idx, torch.int32, 'idx', tuple(vals.shape))
    P, R, k = vals.shape
    ov = torch.empty((R, k), dtype=torch.float32, device=vals.device)
    oi = torch.empty((R, k), dtype=torch.int32, device=vals.device)
    _lib.check(lib.hsk_topk_merge(_p(vals), _p(idx), P, R, k, _p(ov), _p(oi), _stream()), 'hsk_topk_merge')
    return ov, oi


def rank_metrics(topk_idx: torch.Tensor, u_idx: torch.Tensor, label_indptr: torch.Tensor,
                 label_indices: torch.Tensor, ks: Sequence[int]) -> torch.Tensor:
    """-> [R, len(ks), 3] (precision, recall, ndcg) per user and cut-off."""
    _lib.require_gpu()
    lib = _lib.load()
    _chk(topk_idx, torch.int32, 'topk_idx')
    _chk(u_idx, torch.int64, 'u_idx')
    _chk(label_indptr, torch.int64, 'label_indptr')
    _chk(label_indices, torch.int32, 'label_indices')
    R, kmax = topk_idx.shape
    if u_idx.shape != (R,):
        raise ValueError('u_idx / topk_idx row mismatch')
    ks_arr = (ctypes.c_int32 * len(ks))(*[int(x) for x in ks])
    out = torch.empty((R, len(ks), 3), dtype=torch.float32, device=topk_idx.device)
    _lib.check(lib.hsk_rank_metrics(_p(topk_idx), R, kmax, _p(u_idx), label_indptr.numel() - 1, _p(label_indptr),
                                    _p(label_indices), ks_arr,
                                    len(ks), _p(out), _stream()), 'hsk_rank_metrics')
    return out
