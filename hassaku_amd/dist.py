"""Multi-GPU BPR-MF: one process per GPU, RCCL over xGMI through torch.distributed (backend "nccl").

Replaces the reference's single-process nn.DataParallel (train/trainer.py:38-41).  Nothing is replicated:

  * ITEM table RANGE-SHARDED: rank r owns items [item_range(I, r, W)) with their AdamW moments and bias; an item row
    never leaves its owner, neither does its gradient;
  * USER table ROW-SHARDED: rank r owns users u with u % W == r at local row u // W (parameters, moments, the lazy
    AdamW bookkeeping);
  * a step processes a GLOBAL batch of W*B positives (weak scaling: B per GPU).  Every rank draws the same Philox
    stream of negatives for the whole global batch and keeps the (positive, item) entries whose item it owns --
    exactly the samples one GPU would draw for a batch of W*B -- so each rank gathers ~B*(1+N) item rows, like one
    GPU on its own batch.

Per step and rank (C = user slots per owner ~ B + 6 sigma, G = W*B):
    all_gather      user rows    [C, D] -> [W*C, D]   4*D*C bytes to every peer     (owner -> everyone)
    all_reduce      s0           [G]                  positive scores, computed by the positive item's owner
    all_reduce      gsum         [G]                  per positive: sum of its negatives' weights over the ranks
    reduce_scatter  user grads   [W*C, D] -> [C, D]   4*D*C bytes to every peer     (everyone -> owner)
The two scalar reductions are what the BPR loss costs (a positive is coupled to ALL its negatives).  The reference's
other two losses ride the same exchange of rows (nn.DataParallel is loss-agnostic, train/rec_losses.py:27-139):
bce needs NO scalar collective (every logit is weighed on its own); sampled softmax needs one all_gather of the ranks'
(running max, normaliser, s0) triples, [3G] -> [W, 3G], in their place.
The all_gather runs under the preparation (sampling, routing, item sort) of the NEXT batch, which is issued a step
ahead on a side stream; the reduce_scatter runs under the local item-gradient pass + item AdamW.  xGMI is a
point-to-point mesh: all_gather / reduce_scatter use all 7 links of a GPU at once (4*D*C bytes per link).

Evaluation is ITEM-SHARDED the same way (BASELINE configs[3]): per chunk of users the owners all_gather the chunk's
user rows, every rank scores them against its item shard and keeps a local top-k, the (value, id) candidates go to
the rank that merges that user (all_to_all, k*8 bytes per user and rank), metrics are computed there and the per-group
sums are all-reduced at the end.  `gather_item_table()` is the other route the north star names (all_gather of the
item shards: I*D*4 bytes), used for model.pth.
"""
import ctypes
import math
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from hassaku_amd import _lib, hip_ops
from hassaku_amd._lib import HskBprmfShard
from hassaku_amd.hip_ops import ADAM_BETA1, ADAM_BETA2, _chk, _p, _stream


class _Done:
    """Handle of a collective that has already completed (host-staged backends)."""

    def wait(self):
        return True


class Comm:
    """The collectives of the sharded step / evaluation on device tensors.  Backend nccl (= RCCL on ROCm): they run
    on the GPU directly -- `async_op=True` returns the torch Work handle, the collective then runs on RCCL's own
    stream behind everything already queued on the current stream, and `handle.wait()` makes the current stream wait
    for it: kernels launched in between overlap the exchange.  Any other backend (gloo in the tests) is staged through
    host memory, synchronously; the handle is then already complete."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.native = dist.get_backend(group) == 'nccl'

    def _staged(self, t):
        return t.is_cuda and not self.native

    def all_reduce(self, t: torch.Tensor, async_op: bool = False, op: str = 'sum'):
        rop = dist.ReduceOp.MAX if op == 'max' else dist.ReduceOp.SUM
        if not self._staged(t):
            work = dist.all_reduce(t, op=rop, group=self.group, async_op=async_op)
            return work if async_op else None
        h = t.cpu()
        dist.all_reduce(h, op=rop, group=self.group)
        t.copy_(h)
        return _Done() if async_op else None

    def all_gather_into(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out [W*n, ...] = concatenation over the ranks of inp [n, ...]."""
        if not self._staged(inp):
            work = dist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)
            return work if async_op else None
        h = inp.cpu()
        parts = [torch.empty_like(h) for _ in range(self.world)]
        dist.all_gather(parts, h, group=self.group)
        out.copy_(torch.cat(parts, dim=0).view(out.shape))
        return _Done() if async_op else None

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """out [n, ...] = sum over the ranks of their inp[rank*n:(rank+1)*n]."""
        if not self._staged(inp):
            work = dist.reduce_scatter_tensor(out, inp, group=self.group, async_op=async_op)
            return work if async_op else None
        h = inp.cpu()
        dist.all_reduce(h, group=self.group)
        n = out.shape[0]
        out.copy_(h[self.rank * n:(self.rank + 1) * n])
        return _Done() if async_op else None

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        """Equal splits along dim 0: out[j*n:(j+1)*n] on rank i = inp[i*n:(i+1)*n] of rank j."""
        if not self._staged(inp):
            work = dist.all_to_all_single(out, inp, group=self.group, async_op=async_op)
            return work if async_op else None
        h = inp.cpu()
        parts = [torch.empty_like(h) for _ in range(self.world)]
        dist.all_gather(parts, h, group=self.group)
        n = h.shape[0] // self.world
        out.copy_(torch.cat([p[self.rank * n:(self.rank + 1) * n] for p in parts], dim=0))
        return _Done() if async_op else None

    def all_gather(self, inp: torch.Tensor):
        """-> list of world tensors shaped like inp."""
        if not self._staged(inp):
            parts = [torch.empty_like(inp) for _ in range(self.world)]
            dist.all_gather(parts, inp, group=self.group)
            return parts
        h = inp.cpu()
        parts = [torch.empty_like(h) for _ in range(self.world)]
        dist.all_gather(parts, h, group=self.group)
        return [p.to(inp.device) for p in parts]

    def broadcast(self, t: torch.Tensor, src: int = 0):
        if not self._staged(t):
            dist.broadcast(t, src=src, group=self.group)
        else:
            h = t.cpu()
            dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)

    def barrier(self):
        dist.barrier(group=self.group)


class LoopbackComm:
    """ONE rank of a `world`-rank job run on its own, its peers absent -- a measuring device (bench.py `cfg5_shard`:
    rank 3 of 8 of the cfg5 job on one MI355X), not a way to train: the rank's state, shapes and kernels are exactly
    what it has inside the real job, the collectives are local stand-ins.  all_gather: the rank's own segment is put
    in place, the peers' segments keep what the caller put there (stand-in rows); all_reduce: the local contribution
    alone; reduce_scatter: the rank's own segment of its own input.  RCCL time is therefore NOT in what this measures."""
    native = True

    def __init__(self, world: int, rank: int):
        if not (0 <= rank < world):
            raise ValueError('rank outside the world')
        self.world, self.rank, self.group = int(world), int(rank), None

    def all_reduce(self, t, async_op=False, op='sum'):
        return _Done() if async_op else None

    def all_gather_into(self, out, inp, async_op=False):
        n = inp.shape[0]
        out[self.rank * n:(self.rank + 1) * n].copy_(inp)
        return _Done() if async_op else None

    def reduce_scatter(self, out, inp, async_op=False):
        n = out.shape[0]
        out.copy_(inp[self.rank * n:(self.rank + 1) * n])
        return _Done() if async_op else None

    def all_to_all(self, out, inp, async_op=False):
        out.copy_(inp)
        return _Done() if async_op else None

    def all_gather(self, inp):
        return [inp.clone() for _ in range(self.world)]

    def broadcast(self, t, src=0):
        pass

    def barrier(self):
        pass


# ------------------------------------------------------------------------------------------------
# ownership
# ------------------------------------------------------------------------------------------------
def owner_of(u, world: int):
    """(owning rank, local row) of global user id(s) u."""
    return u % world, u // world


def local_user_count(n_users: int, rank: int, world: int) -> int:
    return (n_users - rank + world - 1) // world


def item_range(n_items: int, rank: int, world: int):
    """[lo, hi) of the item ids rank owns (contiguous ranges, sizes differ by at most one)."""
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def user_capacity(owner_share: float, global_batch: int) -> int:
    """User slots per owner rank: mean + 6 sigma of Binomial(global_batch, p) + slack, where p = the largest
    share of the training interactions any rank's users hold (a batch draws its positives from the interactions,
    so heavy users count with their weight)."""
    mean = global_batch * owner_share
    c = int(math.ceil(mean + 6.0 * math.sqrt(max(mean * (1.0 - owner_share), 0.0)) + 8))
    return min(global_batch, (c + 3) // 4 * 4)


def entry_capacity(keep_prob: float, global_batch: int, n_neg: int) -> int:
    """Kept (positive, item) entries per rank and step: the G*N negatives fall into this rank's item range with
    probability keep_prob each (mean + 8 sigma), plus at most G positives."""
    mean = global_batch * n_neg * min(1.0, keep_prob)
    c = int(math.ceil(mean + 8.0 * math.sqrt(mean) + 64)) + global_batch
    return min(global_batch * (n_neg + 1), (c + 3) // 4 * 4)


class ShardedBprMf:
    """Fused BPR-MF AdamW step over `comm.world` GPUs (world == 1 is allowed: same code path, collectives degenerate).

    Construct with the FULL tables (identical on every rank, e.g. from the seeded model init) -- the local shards are
    cut out here and the full tensors are not referenced afterwards -- or, with `inputs_are_shards=True`, with this
    rank's shards directly (user rows rank::world, item rows item_range(...)) plus `n_users` / `n_items`."""

    def __init__(self, comm: Comm, user_emb, item_emb, item_bias=None, user_bias=None, global_bias=None, *, lr, wd,
                 batch, n_neg, csr_indptr, csr_indices, coo_user, coo_item, seed=0, beta1=ADAM_BETA1,
                 beta2=ADAM_BETA2, eps=None, capacity: Optional[int] = None, entry_cap: Optional[int] = None,
                 loss='bpr', log_adjust=0.0, alias=None, optimizer='adamw', lazy_items='auto', prefetch=True,
                 inputs_are_shards=False, n_users: Optional[int] = None, n_items: Optional[int] = None,
                 flush_every: int = 0, item_shard=None, native='auto'):
        """native: who issues the step.  True: ONE C call per step (hsk_shard_step), the collectives on the library's own
        RCCL communicator; False: phase by phase from here, torch.distributed in between; 'host-staged': hsk_shard_step over
        the library's host-staged collectives -- for ranks that are processes sharing one GPU (tests: RCCL refuses two ranks
        on one device); 'auto': HSK_SHARD_NATIVE=1 / 0 if set, else True on a one-rank nccl group and False otherwise --
        the natively issued step at world > 1 is taken only when asked for (bench.py asks after it has checked it bit for
        bit against the phased path on the job's own ranks)."""
        _lib.require_gpu()
        self.lib = _lib.load()
        self.comm = comm
        W, r = comm.world, comm.rank
        if loss not in hip_ops.LOSS_KINDS:
            raise ValueError(f'unknown loss {loss!r}')
        if loss == 'bce' and (user_bias is not None or global_bias is not None):
            raise ValueError('bce sends gradient to the user / global bias: the sharded step does not carry them')
        self.loss = loss
        dev = user_emb.device
        D = user_emb.shape[1]
        if inputs_are_shards:
            if n_users is None or n_items is None:
                raise ValueError('inputs_are_shards=True needs n_users and n_items (the global sizes)')
            U, I = int(n_users), int(n_items)
        else:
            U, I = user_emb.shape[0], item_emb.shape[0]
        lo, hi = item_range(I, r, W)
        if item_shard is not None:
            # this rank's item range given explicitly (inputs_are_shards only): one rank's share of a LARGER job run on
            # its own -- e.g. the range rank 3 of 8 owns of the cfg5 catalogue, with a 1-rank group: it samples over all
            # n_items and keeps what falls into [lo, hi), exactly as that rank would (tests, bench.py cfg5_shard)
            if not inputs_are_shards:
                raise ValueError('item_shard needs inputs_are_shards=True')
            lo, hi = int(item_shard[0]), int(item_shard[1])
            if not (0 <= lo < hi <= I):
                raise ValueError(f'item_shard {item_shard} outside [0, {I})')
        if hi <= lo:
            raise ValueError(f'rank {r} of {W} would own no items (n_items = {I})')
        self.device, self.n_users_global, self.n_items, self.dim = dev, U, I, D
        self.item_lo, self.item_hi = lo, hi
        self.batch, self.n_neg = int(batch), int(n_neg)
        G, K = W * self.batch, self.n_neg + 1
        if inputs_are_shards:
            self.user_emb, self.item_emb = user_emb, item_emb
            self.user_bias = None if user_bias is None else user_bias.reshape(-1)
            self.item_bias = None if item_bias is None else item_bias.reshape(-1)
        else:
            # every rank must start from bit-identical tables: rank 0's copy wins
            for t in (user_emb, item_emb, item_bias, user_bias, global_bias):
                if t is not None and W > 1:
                    comm.broadcast(t, src=0)
            self.user_emb = user_emb[r::W].contiguous()
            self.user_bias = None if user_bias is None else user_bias.reshape(-1)[r::W].contiguous()
            self.item_emb = item_emb[lo:hi].contiguous()
            self.item_bias = None if item_bias is None else item_bias.reshape(-1)[lo:hi].contiguous()
        self.global_bias = global_bias
        U_loc, I_loc = self.user_emb.shape[0], hi - lo
        if U_loc != local_user_count(U, r, W) or self.item_emb.shape[0] != I_loc:
            raise ValueError('shard shapes do not match the ownership rule')
        for t, name in ((csr_indptr, 'csr_indptr'), (coo_user, 'coo_user'), (coo_item, 'coo_item')):
            if t is None:
                raise ValueError(f'{name} is required')
        _chk(csr_indptr, torch.int64, 'csr_indptr', (U + 1,))
        _chk(csr_indices, torch.int32, 'csr_indices')
        _chk(coo_user, torch.int32, 'coo_user')
        _chk(coo_item, torch.int32, 'coo_item', tuple(coo_user.shape))
        self._keep = (csr_indptr, csr_indices, coo_user, coo_item)
        nnz = coo_user.numel()

        # capacities from the data: the share of the interactions the busiest owner holds, and the probability
        # that a drawn negative falls into this rank's item range (interaction-weighted over the users)
        if capacity is None:
            if nnz > (1 << 28):   # (a count over billions of interactions: the shares are 1/W to four digits)
                share = 1.02 / W
            else:
                share = torch.bincount(coo_user.long() % W, minlength=W).double().max().item() / max(nnz, 1)
            capacity = user_capacity(min(1.0, max(share, 1.0 / W)), G)
        if entry_cap is None:
            deg = (csr_indptr[1:] - csr_indptr[:-1]).double()
            inv_free = float((deg / (I - deg).clamp(min=1.0)).sum().item()) / max(float(deg.sum().item()), 1.0)
            keep = I_loc * inv_free
            if alias is not None:     # 'popular' sampling: mass of the item law inside the range, with headroom
                prob, idx = alias[0].double(), alias[1].long()
                mass = torch.zeros(I, dtype=torch.float64, device=dev).index_add_(0, idx, 1.0 - prob) + prob
                keep = min(1.0, 1.5 * float(mass[lo:hi].sum().item()) / I + 0.02)
            entry_cap = entry_capacity(keep, G, self.n_neg)
        C, cap = int(min(capacity, G)), int(min(entry_cap, G * K))
        if W > 1:   # the all_gather / reduce_scatter need one C on every rank
            agreed = torch.tensor([C], dtype=torch.int64, device=dev)
            comm.all_reduce(agreed, op='max')
            C = int(agreed.item())
        self.capacity, self.entry_cap = C, cap
        self.params = dict(user_emb=self.user_emb, item_emb=self.item_emb, item_bias=self.item_bias,
                           user_bias=self.user_bias, global_bias=global_bias)
        self.m = {k: (torch.zeros_like(t) if t is not None else None) for k, t in self.params.items()}
        self.v = {k: (torch.zeros_like(t) if t is not None else None) for k, t in self.params.items()}
        self.loss_out = torch.zeros(2, dtype=torch.float64, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        max_batch, max_cols = max(G, C), K
        # (the sharded carving: no [max_batch, D] row buffers -- the step works through rows_all / dU_all)
        nbytes = self.lib.hsk_shard_base_workspace_bytes(U_loc, I_loc, D, max_batch, max_cols)
        sbytes = self.lib.hsk_shard_workspace_bytes(max_batch, max_cols, C, cap)
        if nbytes <= 0 or sbytes <= 0:
            raise ValueError('invalid workspace request')
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.shard_ws = torch.empty(sbytes, dtype=torch.uint8, device=dev)
        self.rows_send = torch.zeros((C, D), dtype=torch.float32, device=dev)
        self.rows_all = torch.zeros((W * C, D), dtype=torch.float32, device=dev)
        self.dU_all = torch.zeros((W * C, D), dtype=torch.float32, device=dev)
        self.grads_mine = torch.zeros((C, D), dtype=torch.float32, device=dev)
        self.s0 = torch.zeros(G, dtype=torch.float32, device=dev)
        self.gsum = torch.zeros(G, dtype=torch.float32, device=dev)
        # sampled softmax: the ranks' (running max, normaliser, s0) triples travel in one all_gather
        self.ssm_send = torch.zeros(3 * G, dtype=torch.float32, device=dev) if loss == 'sampled_softmax' else None
        self.ssm_all = torch.zeros(W * 3 * G, dtype=torch.float32, device=dev) if loss == 'sampled_softmax' else None

        sh = HskBprmfShard()
        st = sh.base
        for k, t in self.params.items():
            setattr(st, k, _p(t))
            setattr(st, 'm_' + k, _p(self.m[k]))
            setattr(st, 'v_' + k, _p(self.v[k]))
        st.n_users, st.n_items, st.dim = U_loc, I_loc, D
        st.lr, st.beta1, st.beta2, st.eps, st.wd = lr, beta1, beta2, hip_ops.opt_eps(optimizer, eps), wd
        st.opt_kind = hip_ops.OPT_KINDS[optimizer]
        st.step = 0
        st.csr_indptr, st.csr_indices = _p(csr_indptr), _p(csr_indices)
        st.coo_user, st.coo_item, st.nnz = _p(coo_user), _p(coo_item), nnz
        st.seed = seed & 0xFFFFFFFFFFFFFFFF
        st.workspace, st.workspace_bytes = _p(self.workspace), nbytes
        st.max_batch, st.max_cols = max_batch, max_cols
        st.lazy_users = 1
        st.ws_sharded, st.flush_every = 1, int(flush_every)
        st.graph_chunk, st.catchup_apart = -1, 0
        if lazy_items == 'auto':
            # Worth it while a batch leaves enough of the shard's rows alone.  With a fraction f = 1 - exp(-entries / rows)
            # of the rows touched per step the dense sweep moves 6 row-units (p, m, v read and written) per row, the lazy
            # update 6 f for the touched rows + 4 f (1 - f) for the catch-up of those that skipped a step, at a somewhat
            # lower rate per byte (random whole rows instead of a stream): measured break-even f ~ 0.8 (configs[4]: f =
            # 0.73, 7.6 against 7.9 ms per step) -- i.e. entries / rows < 1.5.
            lazy_items = D % 2 == 0 and I_loc >= 0.66 * cap and I_loc * D > hip_ops.LAZY_USERS_MIN_ELEMENTS
        st.lazy_items = 1 if lazy_items else 0
        st.timing_mask, st.timing, st.aux, st.timing_every = 0, None, None, 1
        st.loss_kind, st.ssm_log_adjust = hip_ops.LOSS_KINDS[loss], float(log_adjust)
        self.alias = alias
        st.alias_prob, st.alias_idx = (None, None) if alias is None else (_p(alias[0]), _p(alias[1]))
        st.loss_out, st.status = _p(self.loss_out), _p(self.status)
        sh.world, sh.rank = W, r
        sh.n_users_global, sh.n_items_global, sh.item_lo = U, I, lo
        sh.capacity, sh.entry_cap = C, cap
        sh.shard_ws, sh.shard_ws_bytes = _p(self.shard_ws), sbytes
        sh.rows_send, sh.rows_all = _p(self.rows_send), _p(self.rows_all)
        sh.dU_all, sh.grads_mine = _p(self.dU_all), _p(self.grads_mine)
        sh.s0, sh.gsum = _p(self.s0), _p(self.gsum)
        sh.ssm_send, sh.ssm_all = _p(self.ssm_send), _p(self.ssm_all)
        self.sh = sh
        _lib.check(self.lib.hsk_bprmf_init_workspace(ctypes.byref(sh.base), _stream()), 'hsk_bprmf_init_workspace')
        _lib.check(self.lib.hsk_shard_init(ctypes.byref(sh), _stream()), 'hsk_shard_init')
        # the next batch is prepared a step ahead on a side stream (device-side counterpart of DataLoader prefetch)
        self._side = torch.cuda.Stream(device=dev) if prefetch else None
        self._ev_fork, self._ev_ready = torch.cuda.Event(), torch.cuda.Event()
        self._cur_set = 0
        self._pf = None          # (order ptr, start, batch, step index it is for, set, order tensor kept alive)
        self._prefetch = bool(prefetch)
        # The whole step from ONE C call (csrc/hsk_rccl.inc: hsk_shard_step) -- the phase-by-phase sequence below costs ten
        # Python / torch dispatches per step and leaves the GPU idle in between.  The library opens its own communicator
        # over the same ranks; the 128-byte id travels through this process group.
        self._rt = None
        self._order_keep = None
        import os
        if native == 'auto':
            env = os.environ.get('HSK_SHARD_NATIVE')
            native = (env == '1') if env in ('0', '1') else (W == 1 and isinstance(comm, Comm) and comm.native)
        if native == 'host-staged':
            if not isinstance(comm, Comm):
                raise ValueError("native='host-staged' needs a process group to hand the segment's name around")
            self._open_runtime(lambda buf: self.lib.hsk_hostcoll_unique_id(W * C * D, buf), 'hsk_hostcoll_unique_id')
        elif native:
            if not (isinstance(comm, Comm) and comm.native and self.lib.hsk_rccl_available()):
                raise RuntimeError('native=True needs backend nccl and a loadable librccl')
            self._open_runtime(self.lib.hsk_rccl_unique_id, 'hsk_rccl_unique_id')

    def _open_runtime(self, make_id, what):
        comm, W, r, dev = self.comm, self.comm.world, self.comm.rank, self.device
        idb = torch.zeros(128, dtype=torch.uint8)
        if r == 0:
            buf = (ctypes.c_ubyte * 128)()
            _lib.check(make_id(buf), what)
            idb = torch.tensor(list(buf), dtype=torch.uint8)
        idd = idb.to(dev)
        comm.broadcast(idd, src=0)
        raw = bytes(idd.cpu().tolist())
        rt = self.lib.hsk_shard_rt_create(W, r, ctypes.c_char_p(raw))
        # every rank takes the native path or none does (a rank on its own in the phased path would wait for ever)
        ok = torch.tensor([1.0 if rt else 0.0], device=dev)
        comm.all_reduce(ok)
        if int(ok.item()) == W:
            self._rt = ctypes.c_void_p(rt)
        else:
            import warnings
            warnings.warn('hsk_shard_rt_create failed on a rank (' + self.lib.hsk_last_error().decode('utf-8', 'replace')
                          + '): the sharded step falls back to the phase-by-phase path')
            if rt:
                self.lib.hsk_shard_rt_destroy(ctypes.c_void_p(rt))

    @property
    def issued_natively(self) -> bool:
        return self._rt is not None

    def backend(self) -> str:
        """who moves the bytes: 'rccl' / 'host-staged' (hsk_shard_step) or 'torch.distributed:<backend>' (phased)"""
        if self._rt is not None:
            return self.lib.hsk_shard_rt_backend(self._rt).decode()
        return 'torch.distributed:' + ('nccl' if getattr(self.comm, 'native', False) else 'staged')

    def set_hyper(self, lr=None, wd=None, beta1=None, beta2=None, eps=None):
        """Change lr / wd / betas / eps between steps (an LR schedule) -- on EVERY rank, with the same values.  The
        per-step Adam scalars of the lazy replay are a device table computed from the hyper-parameters, so: flush under
        the old values, install the new ones, prepare the workspace again (train/trainer.py:52-53 of the reference builds
        its optimiser once; torch schedulers mutate param_groups[...]['lr'] the same way)."""
        self.flush()
        torch.cuda.synchronize()
        st = self.sh.base
        for name, val in (('lr', lr), ('wd', wd), ('beta1', beta1), ('beta2', beta2), ('eps', eps)):
            if val is not None:
                setattr(st, name, float(val))
        _lib.check(self.lib.hsk_bprmf_init_workspace(ctypes.byref(st), _stream()), 'hsk_bprmf_init_workspace')
        _lib.check(self.lib.hsk_shard_init(ctypes.byref(self.sh), _stream()), 'hsk_shard_init')
        self._pf = None

    @property
    def step_count(self) -> int:
        return int(self.sh.base.step)

    def flush_cadence(self):
        """(steps between sweeps of the local user shard, of the local item shard); 2**30 = never (explicit flush only)"""
        st = ctypes.byref(self.sh.base)
        touched_items = self.item_emb.shape[0] * (1.0 - math.exp(-self.entry_cap / self.item_emb.shape[0]))
        return (int(self.lib.hsk_bprmf_flush_cadence(st, 0, self.capacity)),
                int(self.lib.hsk_bprmf_flush_cadence(st, 1, max(1, int(touched_items)))))

    # -- one global step -------------------------------------------------------------------------------------
    def _prepare(self, order, start_global, nb, set_, stream):
        _lib.check(self.lib.hsk_shard_prepare(ctypes.byref(self.sh), _p(order), start_global, nb, self.n_neg, set_,
                                              stream), 'hsk_shard_prepare')

    def step_sampled(self, order: Optional[torch.Tensor], start_global: int, batch: Optional[int] = None,
                     next_start: Optional[int] = None, next_batch: Optional[int] = None):
        """One global step: positives = interactions order[start_global : start_global + world*batch].
        `next_start` (and `next_batch`) name the batch of the FOLLOWING call: it is sampled, routed and item-sorted on
        the side stream while this step's exchanges run (results are identical with or without it)."""
        lib, sh, comm = self.lib, self.sh, self.comm
        nb = self.batch if batch is None else int(batch)
        G = comm.world * nb
        if nb > self.batch:
            raise ValueError(f'batch {nb} above the batch {self.batch} the state was built for')
        if order is not None:
            _chk(order, torch.int64, 'order')
            if start_global + G > order.numel():
                raise ValueError('order too short for the global batch')
        main = torch.cuda.current_stream()
        s = main.cuda_stream
        ref = ctypes.byref(sh)
        if self._rt is not None:
            nxt = -1 if (next_start is None or not self._prefetch) else int(next_start)
            _lib.check(lib.hsk_shard_step(ref, self._rt, _p(order), int(start_global), nb, self.n_neg, nxt,
                                          0 if next_batch is None else int(next_batch), s), 'hsk_shard_step')
            self._order_keep = order          # the prepared batch reads it during the next call
            self._cur_set = int(lib.hsk_shard_rt_cur_set(self._rt))
            return
        key = (_p(order), int(start_global), nb, self.step_count)
        if self._pf is not None and self._pf[:4] == key:
            main.wait_event(self._ev_ready)              # sampled + sorted during the previous step
            set_ = self._pf[4]
            self._pf = None
        else:
            self._discard_prefetch()
            set_ = self._cur_set ^ 1
            self._prepare(order, start_global, nb, set_, s)
        self._cur_set = set_
        _lib.check(lib.hsk_shard_pack(ref, nb, self.n_neg, set_, s), 'hsk_shard_pack')
        rows = comm.all_gather_into(self.rows_all, self.rows_send, async_op=True)
        if self._side is not None and next_start is not None:
            nnb = nb if next_batch is None else int(next_batch)
            self._ev_fork.record(main)
            self._side.wait_event(self._ev_fork)
            self._prepare(order, int(next_start), nnb, set_ ^ 1, self._side.cuda_stream)
            self._ev_ready.record(self._side)
            self._pf = (_p(order), int(next_start), nnb, self.step_count + 1, set_ ^ 1, order)
        rows.wait()
        _lib.check(lib.hsk_shard_pos_scores(ref, s), 'hsk_shard_pos_scores')
        if self.loss == 'bpr':                       # (bce: no scalar exchange at all)
            comm.all_reduce(self.s0[:G])
        _lib.check(lib.hsk_shard_forward(ref, s), 'hsk_shard_forward')
        if self.loss == 'bpr':
            comm.all_reduce(self.gsum[:G])
        elif self.loss == 'sampled_softmax':         # the ranks' (max, normaliser, s0) triples: [3G] -> [W, 3G]
            comm.all_gather_into(self.ssm_all[:comm.world * 3 * G], self.ssm_send[:3 * G])
        _lib.check(lib.hsk_shard_pos_fix(ref, s), 'hsk_shard_pos_fix')
        grads = comm.reduce_scatter(self.grads_mine, self.dU_all, async_op=True)
        _lib.check(lib.hsk_shard_apply_items(ref, s), 'hsk_shard_apply_items')     # under the reduce_scatter
        grads.wait()
        _lib.check(lib.hsk_shard_apply_users(ref, s), 'hsk_shard_apply_users')

    def _discard_prefetch(self):
        """A prepared batch that the next call does not consume: give its owner map back."""
        if self._rt is not None:
            _lib.check(self.lib.hsk_shard_rt_discard_prefetch(ctypes.byref(self.sh), self._rt, _stream()),
                       'hsk_shard_rt_discard_prefetch')
            return
        if self._pf is None:
            return
        torch.cuda.current_stream().wait_event(self._ev_ready)
        _lib.check(self.lib.hsk_shard_discard(ctypes.byref(self.sh), self._pf[4], _stream()), 'hsk_shard_discard')
        self._pf = None

    def peek_batch(self, order: Optional[torch.Tensor], start_global: int, batch: Optional[int] = None):
        """The batch the NEXT step_sampled(order, start_global, batch) will train on, as this rank keeps it (same
        triple as last_batch()) -- debug / parity: it is prepared into the idle buffer set, read and discarded; the
        step then prepares it again (same RNG stream id: identical samples)."""
        nb = self.batch if batch is None else int(batch)
        self._discard_prefetch()
        set_ = self._cur_set ^ 1
        self._prepare(order, int(start_global), nb, set_, _stream())
        out = self._read_batch(set_, nb)
        _lib.check(self.lib.hsk_shard_discard(ctypes.byref(self.sh), set_, _stream()), 'hsk_shard_discard')
        return out

    def _read_batch(self, set_, nb):
        G = self.comm.world * nb
        offs = torch.empty(G + 1, dtype=torch.int32, device=self.device)
        items = torch.empty(self.entry_cap, dtype=torch.int32, device=self.device)
        u = torch.empty(G, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.hsk_shard_last_batch(ctypes.byref(self.sh), set_, nb, _p(offs), _p(items), _p(u),
                                                 _stream()), 'hsk_shard_last_batch')
        return offs, items, u

    def last_batch(self, batch: Optional[int] = None):
        """(offs int32 [G+1], local item ids int32 [entry_cap] (-1 beyond the kept entries), users int32 [G]) of the
        batch of the latest step -- debug / parity."""
        return self._read_batch(self._cur_set, self.batch if batch is None else int(batch))

    # -- per-stage device timing (same recorder as the single-GPU state; 'fwd', 'item', 'user' are bracketed) ------
    def enable_timing(self, stages=('fwd',), every=1):
        if not getattr(self, '_timing', None):
            self._timing = self.lib.hsk_timing_create()
        names = hip_ops.BprMfFusedState.STAGES
        self.sh.base.timing = self._timing
        self.sh.base.timing_mask = sum(1 << names.index(s) for s in stages if s in ('fwd', 'item', 'user'))
        self.sh.base.timing_every = int(every)

    def disable_timing(self):
        self.sh.base.timing_mask = 0

    def collect_timing(self):
        if not getattr(self, '_timing', None):
            return {}
        names = hip_ops.BprMfFusedState.STAGES
        ms = (ctypes.c_double * len(names))()
        cnt = (ctypes.c_int64 * len(names))()
        _lib.check(self.lib.hsk_timing_collect(self._timing, ms, cnt), 'hsk_timing_collect')
        return {s: (ms[i], cnt[i]) for i, s in enumerate(names) if cnt[i] > 0}

    def flush(self):
        self._discard_prefetch()
        _lib.check(self.lib.hsk_shard_flush(ctypes.byref(self.sh), _stream()), 'hsk_shard_flush')

    def close(self):
        """Release the library's RCCL communicator and streams (collective in spirit: call it on every rank)."""
        rt, self._rt = getattr(self, '_rt', None), None
        if rt is not None:
            torch.cuda.synchronize()
            self.lib.hsk_shard_rt_destroy(rt)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

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
            raise RuntimeError(f'{what}: a capacity was exceeded (user slots per owner {self.capacity}, kept entries '
                               f'per rank {self.entry_cap}); construct ShardedBprMf with larger `capacity` / `entry_cap`')
        hip_ops.raise_on_status(self.status, what)
        if int(bad.item()):
            raise RuntimeError(f'{what}: another rank reported a failure')

    # -- assembling the full tables (model.pth, hand-over to a single-GPU consumer) --------------------------
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

    def _gather_item_rows(self, t):
        """all_gather of one per-item tensor of the item shards -> its full [I, ...] form."""
        W, I = self.comm.world, self.n_items
        n_max = max(item_range(I, r, W)[1] - item_range(I, r, W)[0] for r in range(W))
        pad = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        out = torch.empty((W * n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self.comm.all_gather_into(out, pad)
        return torch.cat([out[r * n_max: r * n_max + item_range(I, r, W)[1] - item_range(I, r, W)[0]]
                          for r in range(W)], dim=0)

    def gather_item_table(self):
        """Full [I, D] item table (and [I] item bias): the all_gather of the item shards."""
        self.flush()
        return (self._gather_item_rows(self.item_emb),
                None if self.item_bias is None else self._gather_item_rows(self.item_bias))

    def gather_item_moments(self):
        """Full [I, D] exp_avg and exp_avg_sq of the item table (diagnostics: a parameter that differs between two
        summation orders while both moments agree is Adam's division at work, tests/conftest.py)."""
        self.flush()
        return self._gather_item_rows(self.m['item_emb']), self._gather_item_rows(self.v['item_emb'])


class TableShards:
    """This rank's share of a trained model (what the item-sharded evaluation reads): users u % W == r at row u // W,
    items item_range(I, r, W).  ShardedBprMf offers the same attributes."""

    def __init__(self, comm, user_emb, item_emb, item_bias, user_bias, global_bias, n_users, n_items):
        self.comm, self.device = comm, user_emb.device
        self.user_emb, self.item_emb, self.item_bias, self.user_bias = user_emb, item_emb, item_bias, user_bias
        self.global_bias = global_bias
        self.n_users_global, self.n_items, self.dim = int(n_users), int(n_items), user_emb.shape[1]
        self.item_lo, self.item_hi = item_range(self.n_items, comm.rank, comm.world)
        if item_emb.shape[0] != self.item_hi - self.item_lo or \
                user_emb.shape[0] != local_user_count(self.n_users_global, comm.rank, comm.world):
            raise ValueError('shard shapes do not match the ownership rule')

    @staticmethod
    def cut(comm, user_emb, item_emb, item_bias=None, user_bias=None, global_bias=None):
        """Shards of FULL tables that are identical on every rank."""
        W, r = comm.world, comm.rank
        lo, hi = item_range(item_emb.shape[0], r, W)
        return TableShards(comm, user_emb[r::W].contiguous(), item_emb[lo:hi].contiguous(),
                           None if item_bias is None else item_bias.reshape(-1)[lo:hi].contiguous(),
                           None if user_bias is None else user_bias.reshape(-1)[r::W].contiguous(), global_bias,
                           user_emb.shape[0], item_emb.shape[0])

    def flush(self):
        pass


# ------------------------------------------------------------------------------------------------
# item-sharded full evaluation (BASELINE configs[3])
# ------------------------------------------------------------------------------------------------
def evaluate_item_sharded(comm: Comm, sharded, dataset, evaluator, chunk: Optional[int] = None):
    """Full evaluation on the sharded tables (eval/eval.py:237-253 + FullEvaluator of the reference).  Per chunk of
    users: (1) the owners all_gather the chunk's user rows; (2) every rank scores the chunk against ITS item shard --
    `hsk_mf_eval_topk` over the physical slice, exclusion mask restricted to its range -- and keeps a local top-k with
    global item ids; (3) the candidates of user j go to the rank that merges j (all_to_all, k*8 bytes per user and
    rank), which runs `hsk_topk_merge` and the rank metrics for its share of the chunk; (4) per-group sums and counts
    are all-reduced once at the end.  The candidate exchange of a chunk runs under the scoring of the next one.
    `sharded` is a ShardedBprMf or a TableShards; `dataset` a FullEvalDataset (its CSRs are global); `evaluator`
    supplies K_VALUES and the user groups."""
    sharded.flush()
    W, r = comm.world, comm.rank
    dev = sharded.device
    arr = dataset.device_arrays(dev)
    U, I, D = sharded.n_users_global, sharded.n_items, sharded.dim
    lo_i, I_loc = sharded.item_lo, sharded.item_hi - sharded.item_lo
    ks = sorted(evaluator.K_VALUES, reverse=True)
    k = ks[0]
    if I < k:
        raise ValueError(f'full evaluation needs at least {k} items (K_VALUES), got {I}')
    kk = min(k, I_loc)
    if W * k > 4096:
        raise ValueError(f'item-sharded evaluation merges world x k = {W} x {k} candidates per user; hsk_topk_merge takes '
                         f'at most 4096 (use fewer ranks per evaluation group or a smaller K)')
    n_groups = evaluator.get_n_groups()
    groups = evaluator.get_user_to_user_group().to(dev) if n_groups > 0 else None
    sums = torch.zeros((n_groups + 1, len(ks), 3), dtype=torch.float64, device=dev)
    counts = torch.zeros(n_groups + 1, dtype=torch.float64, device=dev)
    status = hip_ops.new_status(dev)
    if chunk is None:        # a rank scores chunk x I/W: 2048*W users per chunk keep its GEMM the size of one GPU's
        chunk = min(2048 * W, U)
    chunk = max(W, (int(chunk) + W - 1) // W * W)         # chunk boundaries at multiples of W: a rank's rows of a chunk
    S = chunk // W                                         # are a contiguous slice of its shard; S users merged per rank
    send_u = torch.zeros((S, D), dtype=torch.float32, device=dev)
    all_u = torch.empty((W * S, D), dtype=torch.float32, device=dev)
    has_ub = sharded.user_bias is not None
    send_b = torch.zeros(S, dtype=torch.float32, device=dev) if has_ub else None
    all_b = torch.empty(W * S, dtype=torch.float32, device=dev) if has_ub else None
    bufs = [dict(inp=torch.empty((W * S, 2 * k), dtype=torch.int32, device=dev),
                 out=torch.empty((W * S, 2 * k), dtype=torch.int32, device=dev)) for _ in range(2)]
    pending = None

    def finish(p):
        work, out, lo, n = p
        work.wait()
        c = out.view(W, S, 2 * k)
        cand_v = c[:, :, :k].contiguous().view(torch.float32)
        cand_i = c[:, :, k:].contiguous()
        _, ids = hip_ops.topk_merge(cand_v, cand_i)
        n_mine = max(0, min(S, n - r * S))                 # this rank merges users lo + r*S .. of the chunk
        if n_mine == 0:
            return
        um = torch.arange(lo + r * S, lo + r * S + n_mine, device=dev)
        met = hip_ops.rank_metrics(ids[:n_mine].contiguous(), um, arr['label_indptr'], arr['label_indices'], ks).double()
        sums[0] += met.sum(0)
        counts[0] += n_mine
        for g in range(n_groups):
            sel = groups[um] == g
            sums[1 + g] += (met * sel.view(-1, 1, 1)).sum(0)
            counts[1 + g] += sel.sum()

    with torch.no_grad():
        for ci, lo in enumerate(range(0, U, chunk)):
            n = min(chunk, U - lo)
            # (1) user rows of the chunk: global user lo + j lives on rank j % W at local row (lo + j) // W
            mine = local_user_count(n, r, W)               # users lo + r, lo + r + W, ... < lo + n
            send_u.zero_()
            send_u[:mine] = sharded.user_emb[lo // W: lo // W + mine]
            comm.all_gather_into(all_u, send_u)
            cu = all_u.view(W, S, D).transpose(0, 1).reshape(W * S, D)[:n].contiguous()
            cb = None
            if has_ub:
                send_b.zero_()
                send_b[:mine] = sharded.user_bias[lo // W: lo // W + mine]
                comm.all_gather_into(all_b, send_b)
                cb = all_b.view(W, S).t().reshape(W * S)[:n].contiguous()
            # (2) local scores + top-k over the physical item shard
            u = torch.arange(n, device=dev)
            v, i, _ = hip_ops.mf_eval_topk(cu, sharded.item_emb, sharded.item_bias, cb, sharded.global_bias, u, kk,
                                           arr['excl_indptr'][lo: lo + n + 1], arr['excl_indices'], item_begin=lo_i,
                                           item_count=I_loc, status=status, item_shard=True,
                                           n_items_global=I)
            b = bufs[ci & 1]
            inp = b['inp'].view(W * S, 2, k)
            inp[:, 0, :] = torch.tensor(float('-inf'), device=dev).view(torch.int32)
            inp[:, 1, :] = 2 ** 31 - 1
            inp[:n, 0, :kk] = v.view(torch.int32)
            inp[:n, 1, :kk] = i
            # (3) candidates to the merging rank; waited for after the next chunk has been issued
            work = comm.all_to_all(b['out'], b['inp'], async_op=True)
            if pending is not None:
                finish(pending)
            pending = (work, b['out'], lo, n)
        if pending is not None:
            finish(pending)
    hip_ops.raise_on_status(status, 'item-sharded eval')
    comm.all_reduce(sums)
    comm.all_reduce(counts)
    return _metric_dict(sums.cpu(), counts.cpu(), ks, n_groups)


def init_shard_tables(rank: int, world: int, n_users: int, n_items: int, dim: int, device, seed: int = 64,
                      item_bias: bool = True, user_bias: bool = False, rows_per_call: int = 1 << 20):
    """This rank's shards of freshly initialised tables, created directly on the device: user rows rank::world, item
    rows item_range(...).  The law is the reference's (general_weight_init, train/utils.py:5-13: N(0, (0.1/D)^2) for the
    embeddings, N(0, 0.1^2) for the [., 1] bias tables); the STREAM is per shard (seed, rank) -- tables this large never
    exist in one piece, so there is no single-device initialisation to be bit-identical to (smaller jobs keep the
    seeded full-table init and cut it: ShardedBprMf without inputs_are_shards).  -> dict of tensors."""
    device = torch.device(device)
    U_loc = local_user_count(n_users, rank, world)
    lo, hi = item_range(n_items, rank, world)
    gen = torch.Generator(device=device)
    gen.manual_seed((int(seed) * 1000003 + rank) & 0x7fffffffffffffff)

    def table(rows, cols, std):
        t = torch.empty((rows, cols), dtype=torch.float32, device=device)
        for r0 in range(0, rows, rows_per_call):       # bounded calls: a 51 GB table is 12.8e9 elements
            t[r0:r0 + rows_per_call].normal_(mean=0.0, std=std, generator=gen)
        return t

    out = {'user_emb': table(U_loc, dim, 0.1 / dim), 'item_emb': table(hi - lo, dim, 0.1 / dim),
           'item_bias': table(hi - lo, 1, 0.1).view(-1) if item_bias else None,
           'user_bias': table(U_loc, 1, 0.1).view(-1) if user_bias else None}
    return out


def _metric_dict(sums, counts, ks, n_groups):
    out = {}
    for gi in range(n_groups + 1):
        prefix = '' if gi == 0 else f'group_{gi - 1}_'
        for t, k in enumerate(ks):
            for j, name in enumerate(('precision', 'recall', 'ndcg')):
                out[f'{prefix}{name}@{k}'] = float(sums[gi, t, j] / counts[gi])
    return out
