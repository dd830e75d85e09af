"""Multi-GPU BPR-MF: one process per GPU, RCCL over xGMI through torch.distributed (backend "nccl").

Replaces the reference's single-process nn.DataParallel (train/trainer.py:38-41).  Layout:

  * user tables ROW-SHARDED: rank r owns users u with u % world == r at local row u // world (parameters,
    AdamW moments, lazy-update bookkeeping all live only on the owner);
  * item table REPLICATED (every item row is touched every step at the BASELINE shapes, so each rank applies the
    same all-reduced dense item gradient);
  * the global batch of world*B positives is cut into contiguous slices, the sampler's RNG is keyed by the global
    batch position, so N GPUs compute the step one GPU would compute on the same world*B batch.

Per step and rank: all_to_all 4*D*C*world B (user rows to the requesters; the requests themselves are recomputed
by the owner, every rank knows every slice of the global batch), all_to_all 4*D*C*world B (user-row gradients back
to the owners), all_reduce 4*(D+1)*I B (item gradient); C = per-pair slot capacity ~ B/world + 6 sigma.  Each
collective is issued asynchronously and has independent kernels running under it: the item sort under the row
exchange, the item-gradient pass under the gradient-row exchange, the user update under the all_reduce.  xGMI is a
point-to-point mesh: the all_to_alls use every link at once; the all_reduce is the bandwidth term (21.9 MB at
I=10 677, D=512).

Evaluation shards the USERS the same way (each rank scores the users it owns against the replicated item
table; per-group metric sums and counts are all-reduced), which needs no table exchange at all.
"""
import ctypes
import math
from typing import Optional

import torch
import torch.distributed as dist

from hassaku_amd import _lib, hip_ops
from hassaku_amd._lib import HskBprmfMp
from hassaku_amd.hip_ops import ADAM_BETA1, ADAM_BETA2, ADAM_EPS, _chk, _p, _stream


class _Done:
    """Handle of a collective that has already completed (host-staged backends)."""

    def wait(self):
        return True


class Comm:
    """The collectives the sharded step needs, on device tensors.  With backend nccl (= RCCL on ROCm) they run on
    the GPU directly -- `async_op=True` returns the torch Work handle, the collective then runs on RCCL's own stream
    behind everything already queued on the current stream, and `handle.wait()` makes the current stream wait for
    it: kernels launched in between overlap the exchange.  Any other backend (gloo in the tests) is staged through
    host memory, synchronously; the handle is then already complete."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.native = dist.get_backend(group) == 'nccl'

    def all_reduce(self, t: torch.Tensor, async_op: bool = False):
        if self.native:
            work = dist.all_reduce(t, group=self.group, async_op=async_op)
            return work if async_op else None
        if not t.is_cuda:
            dist.all_reduce(t, group=self.group)
        else:
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        return _Done() if async_op else None

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """Equal splits along dim 0: out[j*n:(j+1)*n] on rank i = inp[i*n:(i+1)*n] of rank j."""
        if self.native:
            work = dist.all_to_all_single(out, inp, group=self.group, async_op=async_op)
            return work if async_op else None
        h = inp.cpu()
        parts = [torch.empty_like(h) for _ in range(self.world)]
        dist.all_gather(parts, h, group=self.group)
        n = h.shape[0] // self.world
        res = torch.cat([p[self.rank * n:(self.rank + 1) * n] for p in parts], dim=0)
        out.copy_(res)
        return _Done() if async_op else None

    def all_gather(self, inp: torch.Tensor):
        """-> list of world tensors shaped like inp."""
        if self.native or not inp.is_cuda:
            parts = [torch.empty_like(inp) for _ in range(self.world)]
            dist.all_gather(parts, inp, group=self.group)
            return parts
        h = inp.cpu()
        parts = [torch.empty_like(h) for _ in range(self.world)]
        dist.all_gather(parts, h, group=self.group)
        return [p.to(inp.device) for p in parts]

    def broadcast(self, t: torch.Tensor, src: int = 0):
        if self.native or not t.is_cuda:
            dist.broadcast(t, src=src, group=self.group)
        else:
            h = t.cpu()
            dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)

    def barrier(self):
        dist.barrier(group=self.group)


def owner_of(u, world: int):
    """(owning rank, local row) of global user id(s) u."""
    return u % world, u // world


def local_user_count(n_users: int, rank: int, world: int) -> int:
    return (n_users - rank + world - 1) // world


def pair_capacity(batch: int, world: int) -> int:
    """Slots per (source, destination) rank pair: mean + 6 sigma of a Binomial(batch, 1/world), + slack."""
    mean = batch / world
    c = int(math.ceil(mean + 6.0 * math.sqrt(mean * (1.0 - 1.0 / world)) + 8))
    return min(batch, (c + 3) // 4 * 4)


class ShardedBprMf:
    """Fused BPR-MF AdamW step over `comm.world` GPUs.  Construct with the FULL user table (identical on every
    rank, e.g. from the seeded model init); the local shard is cut out here."""

    def __init__(self, comm: Comm, user_emb, item_emb, item_bias=None, user_bias=None, global_bias=None, *, lr, wd,
                 batch, n_neg, csr_indptr, csr_indices, coo_user, coo_item, seed=0, beta1=ADAM_BETA1,
                 beta2=ADAM_BETA2, eps=None, capacity: Optional[int] = None, loss='bpr', log_adjust=0.0, alias=None,
                 optimizer='adamw'):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.comm = comm
        W, r = comm.world, comm.rank
        if W < 2:
            raise ValueError('ShardedBprMf needs world >= 2 (use BprMfFusedState on one GPU)')
        U, D = user_emb.shape
        I = item_emb.shape[0]
        dev = user_emb.device
        self.device, self.n_users_global, self.n_items, self.dim = dev, U, I, D
        self.batch, self.n_neg = int(batch), int(n_neg)
        self.capacity = C = int(capacity or pair_capacity(self.batch, W))
        # the replicated tables must start bit-identical on every rank: rank 0's copy wins
        for t in (item_emb, item_bias, global_bias):
            if t is not None:
                comm.broadcast(t, src=0)
        # local shards (own storage: the full table is not referenced afterwards)
        self.user_emb = user_emb[r::W].contiguous()
        self.user_bias = None if user_bias is None else user_bias.reshape(-1)[r::W].contiguous()
        self.item_emb, self.item_bias, self.global_bias = item_emb, item_bias, global_bias
        U_loc = self.user_emb.shape[0]
        assert U_loc == local_user_count(U, r, W)
        self.params = dict(user_emb=self.user_emb, item_emb=item_emb, item_bias=item_bias, user_bias=self.user_bias,
                           global_bias=global_bias)
        self.m = {k: (torch.zeros_like(t) if t is not None else None) for k, t in self.params.items()}
        self.v = {k: (torch.zeros_like(t) if t is not None else None) for k, t in self.params.items()}
        self.loss_out = torch.zeros(2, dtype=torch.float64, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        R = W * C
        max_batch, max_cols = max(self.batch, R), self.n_neg + 1
        nbytes = self.lib.hsk_bprmf_workspace_bytes(U_loc, I, D, max_batch, max_cols)
        if nbytes <= 0:
            raise ValueError('invalid workspace request')
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.req_send = torch.empty(R, dtype=torch.int32, device=dev)
        self.req_recv = torch.full((R,), -1, dtype=torch.int32, device=dev)
        self.rows_send = torch.zeros((R, D), dtype=torch.float32, device=dev)
        self.rows_recv = torch.zeros((R, D), dtype=torch.float32, device=dev)
        self.grads_send = torch.zeros((R, D), dtype=torch.float32, device=dev)
        self.grads_recv = torch.zeros((R, D), dtype=torch.float32, device=dev)
        # item gradient + item-bias gradient in ONE buffer -> one all_reduce
        self.g_item = torch.zeros(I * D + I, dtype=torch.float32, device=dev)
        self.slot_of_b = torch.zeros(self.batch, dtype=torch.int32, device=dev)
        for t, name in ((csr_indptr, 'csr_indptr'), (coo_user, 'coo_user'), (coo_item, 'coo_item')):
            if t is None:
                raise ValueError(f'{name} is required')
        _chk(csr_indptr, torch.int64, 'csr_indptr', (U + 1,))
        _chk(csr_indices, torch.int32, 'csr_indices')
        _chk(coo_user, torch.int32, 'coo_user')
        _chk(coo_item, torch.int32, 'coo_item', tuple(coo_user.shape))
        self._keep = (csr_indptr, csr_indices, coo_user, coo_item)

        mp = HskBprmfMp()
        st = mp.base
        for k, t in self.params.items():
            setattr(st, k, _p(t))
            setattr(st, 'm_' + k, _p(self.m[k]))
            setattr(st, 'v_' + k, _p(self.v[k]))
        st.n_users, st.n_items, st.dim = U_loc, I, D
        st.lr, st.beta1, st.beta2, st.eps, st.wd = lr, beta1, beta2, hip_ops.opt_eps(optimizer, eps), wd
        st.opt_kind = hip_ops.OPT_KINDS[optimizer]
        st.step = 0
        st.csr_indptr, st.csr_indices = _p(csr_indptr), _p(csr_indices)
        st.coo_user, st.coo_item, st.nnz = _p(coo_user), _p(coo_item), coo_user.numel()
        st.seed = seed & 0xFFFFFFFFFFFFFFFF
        st.workspace, st.workspace_bytes = _p(self.workspace), nbytes
        st.max_batch, st.max_cols = max_batch, max_cols
        st.lazy_users = 1
        st.timing_mask, st.timing, st.aux, st.timing_every, st.timing_now = 0, None, None, 1, 0
        if loss == 'bce' and (user_bias is not None or global_bias is not None):
            raise ValueError('the fused bce step treats user/global bias as gradient-free')
        st.loss_kind, st.ssm_log_adjust = hip_ops.LOSS_KINDS[loss], float(log_adjust)
        self.alias = alias
        st.alias_prob, st.alias_idx = (None, None) if alias is None else (_p(alias[0]), _p(alias[1]))
        st.loss_out, st.status = _p(self.loss_out), _p(self.status)
        mp.world, mp.rank = W, r
        mp.n_users_global, mp.capacity = U, C
        mp.req_send, mp.req_recv = _p(self.req_send), _p(self.req_recv)
        mp.rows_send, mp.rows_recv = _p(self.rows_send), _p(self.rows_recv)
        mp.grads_send, mp.grads_recv = _p(self.grads_send), _p(self.grads_recv)
        mp.g_item_emb = _p(self.g_item)
        mp.g_item_bias = (self.g_item.data_ptr() + 4 * I * D) if item_bias is not None else None
        mp.slot_of_b = _p(self.slot_of_b)
        mp.cur_batch = mp.cur_cols = mp.users_applied = 0
        self.mp = mp
        _lib.check(self.lib.hsk_bprmf_init_workspace(ctypes.byref(mp.base), _stream()), 'hsk_bprmf_init_workspace')

    @property
    def step_count(self) -> int:
        return int(self.mp.base.step)

    def step_sampled(self, order: Optional[torch.Tensor], start_global: int, batch: Optional[int] = None):
        """One global step: positives = interactions order[start_global : start_global + world*batch]."""
        lib, mp, comm = self.lib, self.mp, self.comm
        nb = self.batch if batch is None else int(batch)
        if order is not None:
            _chk(order, torch.int64, 'order')
            if start_global + comm.world * nb > order.numel():
                raise ValueError('order too short for the global batch')
        s = _stream()
        ref = ctypes.byref(mp)
        # sample + route (own requests and, recomputed locally, the incoming ones) + owner catch-up + pack
        _lib.check(lib.hsk_mp_prep(ref, _p(order), start_global, nb, self.n_neg, s), 'hsk_mp_prep')
        rows = comm.all_to_all(self.rows_recv, self.rows_send, async_op=True)
        _lib.check(lib.hsk_mp_sort(ref, s), 'hsk_mp_sort')                    # under the row exchange
        rows.wait()
        _lib.check(lib.hsk_mp_forward(ref, s), 'hsk_mp_forward')
        grads = comm.all_to_all(self.grads_recv, self.grads_send, async_op=True)
        _lib.check(lib.hsk_mp_item_grad(ref, s), 'hsk_mp_item_grad')          # under the gradient-row exchange
        items = comm.all_reduce(self.g_item, async_op=True)
        grads.wait()
        _lib.check(lib.hsk_mp_apply_users(ref, s), 'hsk_mp_apply_users')      # under the item-gradient all_reduce
        items.wait()
        _lib.check(lib.hsk_mp_apply_items(ref, s), 'hsk_mp_apply_items')

    # -- per-stage device timing (same recorder as the single-GPU state; 'fwd' and 'item' are bracketed) -----
    def enable_timing(self, stages=('fwd',), every=1):
        if not getattr(self, '_timing', None):
            self._timing = self.lib.hsk_timing_create()
        names = hip_ops.BprMfFusedState.STAGES
        self.mp.base.timing = self._timing
        self.mp.base.timing_mask = sum(1 << names.index(s) for s in stages)
        self.mp.base.timing_every = int(every)

    def disable_timing(self):
        self.mp.base.timing_mask = 0

    def collect_timing(self):
        if not getattr(self, '_timing', None):
            return {}
        names = hip_ops.BprMfFusedState.STAGES
        ms = (ctypes.c_double * len(names))()
        cnt = (ctypes.c_int64 * len(names))()
        _lib.check(self.lib.hsk_timing_collect(self._timing, ms, cnt), 'hsk_timing_collect')
        return {s: (ms[i], cnt[i]) for i, s in enumerate(names) if cnt[i] > 0}

    def flush(self):
        _lib.check(self.lib.hsk_mp_flush(ctypes.byref(self.mp), _stream()), 'hsk_mp_flush')

    def last_loss(self) -> float:
        t = self.loss_out[:1].clone()
        self.comm.all_reduce(t)
        return float(t.item())

    def pop_loss_sum(self) -> float:
        t = self.loss_out[1:2].clone()
        self.comm.all_reduce(t)
        self.loss_out[1].zero_()
        return float(t.item())

    def check_status(self, what='sharded BPR-MF step'):
        bad = (self.status != 0).to(torch.int32)
        self.comm.all_reduce(bad)        # a flag on any rank fails every rank
        s = int(self.status.item())
        if s & 4:
            raise RuntimeError(f'{what}: request routing overflowed capacity {self.capacity}; '
                               f'construct ShardedBprMf with a larger `capacity`')
        hip_ops.raise_on_status(self.status, what)
        if int(bad.item()):
            raise RuntimeError(f'{what}: another rank reported a failure')

    def gather_user_table(self):
        """Full [U, D] user table (and [U] user bias) assembled from the shards, on every rank."""
        self.flush()
        W, U = self.comm.world, self.n_users_global
        n_max = local_user_count(U, 0, W)

        def gather(t):
            pad = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[:t.shape[0]] = t
            parts = self.comm.all_gather(pad)
            full = torch.empty((U,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            for r, p in enumerate(parts):
                full[r::W] = p[:local_user_count(U, r, W)]
            return full

        return gather(self.user_emb), (None if self.user_bias is None else gather(self.user_bias))


def evaluate_item_sharded(comm: Comm, user_emb, item_emb, item_bias, user_bias, global_bias, dataset, evaluator,
                          chunk: int = 1024):
    """ITEM-sharded full evaluation (BASELINE config 4): rank r scores every user against the item range it owns
    (`hsk_mf_eval_topk` with item_begin/item_count), keeps its local top-k, the (value, id) candidate lists are
    all-gathered (k*8 bytes per user and rank) and merged to the global top-k (`hsk_topk_merge`); metrics are then
    computed for this rank's slice of the users and all-reduced.  Tables are passed whole (replicated) here; a rank
    only reads its item rows, so the same code serves a table that is physically sharded by item range.
    `user_emb` is the FULL user table (e.g. `ShardedBprMf.gather_user_table()`)."""
    W, r = comm.world, comm.rank
    dev = user_emb.device
    arr = dataset.device_arrays(dev)
    n_users, n_items = user_emb.shape[0], item_emb.shape[0]
    lo_i = (n_items * r) // W
    hi_i = (n_items * (r + 1)) // W
    ks = sorted(evaluator.K_VALUES, reverse=True)
    k = ks[0]
    kk = min(k, hi_i - lo_i)
    n_groups = evaluator.get_n_groups()
    groups = evaluator.get_user_to_user_group().to(dev) if n_groups > 0 else None
    sums = torch.zeros((n_groups + 1, len(ks), 3), dtype=torch.float64, device=dev)
    counts = torch.zeros(n_groups + 1, dtype=torch.float64, device=dev)
    status = hip_ops.new_status(dev)
    with torch.no_grad():
        for lo in range(0, n_users, chunk):
            u = torch.arange(lo, min(lo + chunk, n_users), device=dev)
            v, i, _ = hip_ops.mf_eval_topk(user_emb, item_emb, item_bias, user_bias, global_bias, u, kk,
                                           arr['excl_indptr'], arr['excl_indices'], item_begin=lo_i,
                                           item_count=hi_i - lo_i, status=status)
            pv = torch.full((len(u), k), float('-inf'), device=dev)
            pi = torch.full((len(u), k), 2 ** 31 - 1, dtype=torch.int32, device=dev)
            pv[:, :kk], pi[:, :kk] = v, i
            cand_v = torch.stack(comm.all_gather(pv)).contiguous()      # [W, R, k]
            cand_i = torch.stack(comm.all_gather(pi)).contiguous()
            _, ids = hip_ops.topk_merge(cand_v, cand_i)
            mine = slice(r, len(u), W)                                   # metrics: this rank's share of the chunk
            um = u[mine].contiguous()
            met = hip_ops.rank_metrics(ids[mine].contiguous(), um, arr['label_indptr'], arr['label_indices'], ks).double()
            sums[0] += met.sum(0)
            counts[0] += len(um)
            for g in range(n_groups):
                sel = groups[um] == g
                sums[1 + g] += met[sel].sum(0)
                counts[1 + g] += sel.sum()
    hip_ops.raise_on_status(status, 'item-sharded eval')
    comm.all_reduce(sums)
    comm.all_reduce(counts)
    return _metric_dict(sums.cpu(), counts.cpu(), ks, n_groups)


def _metric_dict(sums, counts, ks, n_groups):
    out = {}
    for gi in range(n_groups + 1):
        prefix = '' if gi == 0 else f'group_{gi - 1}_'
        for t, k in enumerate(ks):
            for j, name in enumerate(('precision', 'recall', 'ndcg')):
                out[f'{prefix}{name}@{k}'] = float(sums[gi, t, j] / counts[gi])
    return out


def evaluate_sharded(comm: Comm, sharded: ShardedBprMf, dataset, evaluator, chunk: int = 1024):
    """Users-sharded full evaluation: this rank scores the users it owns; sums and counts are all-reduced.
    `dataset` is a FullEvalDataset; `evaluator` a FullEvaluator (only its K_VALUES / group map are used)."""
    sharded.flush()
    W, r = comm.world, comm.rank
    dev = sharded.device
    key = f'shard{r}of{W}:{dev}'
    cache = dataset._device_cache
    if key not in cache:
        lab, exc = dataset.label_csr.subset_rows(r, W), dataset.exclude_csr.subset_rows(r, W)
        lp, li = lab.to_device(dev)
        ep, ei = exc.to_device(dev)
        cache[key] = dict(label_indptr=lp, label_indices=li, excl_indptr=ep, excl_indices=ei)
    arr = cache[key]
    ks = sorted(evaluator.K_VALUES, reverse=True)
    n_local = sharded.user_emb.shape[0]
    n_groups = evaluator.get_n_groups()
    groups = None
    if n_groups > 0:
        groups = evaluator.get_user_to_user_group().to(dev)[r::W]
    sums = torch.zeros((n_groups + 1, len(ks), 3), dtype=torch.float64, device=dev)
    counts = torch.zeros(n_groups + 1, dtype=torch.float64, device=dev)
    status = hip_ops.new_status(dev)
    with torch.no_grad():
        for lo in range(0, n_local, chunk):
            u = torch.arange(lo, min(lo + chunk, n_local), device=dev)
            _, ids, _ = hip_ops.mf_eval_topk(sharded.user_emb, sharded.item_emb, sharded.item_bias, sharded.user_bias,
                                             sharded.global_bias, u, ks[0], arr['excl_indptr'], arr['excl_indices'],
                                             status=status)
            met = hip_ops.rank_metrics(ids, u, arr['label_indptr'], arr['label_indices'], ks).double()
            sums[0] += met.sum(0)
            counts[0] += len(u)
            for g in range(n_groups):
                sel = groups[u] == g
                sums[1 + g] += met[sel].sum(0)
                counts[1 + g] += sel.sum()
    hip_ops.raise_on_status(status, 'sharded eval')
    comm.all_reduce(sums)
    comm.all_reduce(counts)
    sums, counts = sums.cpu(), counts.cpu()
    out = {}
    for gi in range(n_groups + 1):
        prefix = '' if gi == 0 else f'group_{gi - 1}_'
        for t, k in enumerate(ks):
            for j, name in enumerate(('precision', 'recall', 'ndcg')):
                out[f'{prefix}{name}@{k}'] = float(sums[gi, t, j] / counts[gi])
    return out
