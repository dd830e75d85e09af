"""Trainer -- drop-in for train/trainer.py:15-200 of the reference (same constructor, fit(), val(), log keys).

What changes underneath:
  * No nn.DataParallel and no per-step host synchronisation.  The reference calls `.item()` three times per
    step (train/trainer.py:141-144) only to build per-epoch averages; here the per-step losses are summed on
    the device and read once per epoch.
  * When the model is the HIP SGDMatrixFactorization, the loss is BPR and the optimizer is AdamW, a whole
    step (sampling, gathers, scores, loss, gradients, AdamW on every table) is ONE call into
    libhassaku_hip.so (`hsk_bprmf_train_step[_sampled]`).  Set conf['fused_step'] = False to run the same
    arithmetic through the un-fused operators + torch.optim (autograd path), e.g. for adam / adagrad.
"""
import logging
from typing import Optional

import torch
from tqdm import trange

from hassaku_amd import hip_ops
from hassaku_amd.algorithms.base_classes import SGDBasedRecommenderAlgorithm
from hassaku_amd.algorithms.sgd_alg import SGDMatrixFactorization
from hassaku_amd.data.dataloader import TrainDataLoader
from hassaku_amd.eval.eval import FullEvaluator, evaluate_recommender_algorithm
from hassaku_amd.train.optim import HipOptimizer
from hassaku_amd.train.rec_losses import (RecBayesianPersonalizedRankingLoss, RecBinaryCrossEntropy,
                                          RecommenderSystemLoss, RecSampledSoftmaxLoss)



def _optional_module(name):
    try:
        return __import__(name)
    except ImportError:
        return None


class Trainer:
    def __init__(self, model: SGDBasedRecommenderAlgorithm, train_loader, val_loader, rec_loss: RecommenderSystemLoss,
                 conf: dict):
        self.train_loader, self.val_loader = train_loader, val_loader
        self.device = conf['device']
        if self.device != 'cuda':
            raise RuntimeError("hassaku_amd trains on the HIP device only: set `device: cuda` in the conf "
                               "(ROCm PyTorch names the MI355X 'cuda'); there is no CPU trainer")
        hip_ops._lib.require_gpu()
        import torch.distributed as tdist
        multi = tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
        # one process per GPU: the model arrives with its seeded, full tables in HOST memory (as the reference builds
        # it); only this rank's shards ever go to the device (_build_sharded), so no GPU holds a full table
        self.model = self.pointer_to_model = model if multi else model.to(self.device)
        self.rec_loss = rec_loss
        self.lr, self.wd = conf['lr'], conf['wd']
        if conf['optimizer'] not in hip_ops.OPT_KINDS:
            raise ValueError(f"Optimizer {conf['optimizer']} not yet implemented")

        self.fused: Optional[hip_ops.BprMfFusedState] = None
        self.sharded = None   # hassaku_amd.dist.ShardedBprMf when launched with one process per GPU
        self.comm = None
        want_fused = conf.get('fused_step', True)
        fusable = (isinstance(model, SGDMatrixFactorization) and conf['optimizer'] in hip_ops.OPT_KINDS
                   and isinstance(rec_loss, (RecBayesianPersonalizedRankingLoss, RecBinaryCrossEntropy,
                                             RecSampledSoftmaxLoss)))
        if fusable and isinstance(rec_loss, RecBinaryCrossEntropy) and (model.use_user_bias or model.use_global_bias):
            fusable = False   # bce does send gradient to the user / global bias: autograd path
        if multi:
            # (bce with a user / global bias sends gradient to them: `fusable` is False then, as on one GPU)
            if not (fusable and isinstance(train_loader, TrainDataLoader)):
                raise RuntimeError('multi-GPU training supports mf + {bpr, bce, sampled_softmax} + {adamw, adam, adagrad} '
                                   'with the device TrainDataLoader')
            self.sharded = self._build_sharded(conf)
            self.optimizer = None
        elif want_fused and fusable:
            self.fused = self._build_fused(conf)
            self.pointer_to_model._pre_save_hook = self.fused.flush
            self.optimizer = None
        else:
            # autograd path (any SGD model / loss): forward, loss and backward are HIP autograd functions, and the
            # optimiser step is the same HIP arithmetic as the fused step's (hsk_opt_dense), not torch.optim
            self.optimizer = HipOptimizer(self.model.parameters(), conf['optimizer'], lr=self.lr,
                                          weight_decay=self.wd)

        self.n_epochs = conf['n_epochs']
        self.optimizing_metric = conf['optimizing_metric']
        self.max_patience = conf['max_patience']
        self.model_path = conf['model_path']
        running = conf['running_settings']
        self.use_wandb, self.batch_verbose = running['use_wandb'], running['batch_verbose']
        self._in_tune = conf.get('_in_tune', False)
        self.best_value = self.best_metrics = self.best_epoch = None
        logging.info('Built Trainer: epochs=%d loss=%s fused=%s optimizer=%s lr=%g wd=%g', self.n_epochs,
                     rec_loss.name, self.fused is not None, conf['optimizer'], self.lr, self.wd)

    # ------------------------------------------------------------------------------------------
    def _build_fused(self, conf) -> hip_ops.BprMfFusedState:
        user_emb, item_emb, ib, ub, gb = self.model.tables()
        loader = self.train_loader
        kw = {}
        if isinstance(loader, TrainDataLoader):
            kw = loader.dataset.device_arrays(torch.device(self.device))
            max_batch, n_neg, seed = loader.batch_size, loader.interaction_sampler.n_neg, loader.seed
            kw['alias'] = loader.interaction_sampler.alias(torch.device(self.device))
        else:
            max_batch, n_neg = conf['train_batch_size'], conf['neg_train']
            seed = conf['running_settings'].get('seed', 64)
        return hip_ops.BprMfFusedState(user_emb, item_emb, ib, ub, gb, lr=self.lr, wd=self.wd, max_batch=max_batch,
                                       max_cols=n_neg + 1, seed=seed, loss=self.rec_loss.kind,
                                       log_adjust=getattr(self.rec_loss, 'log_adjust', 0.0),
                                       optimizer=conf['optimizer'], **kw)

    def _build_sharded(self, conf):
        from hassaku_amd.dist import Comm, ShardedBprMf, item_range
        self.comm = Comm()
        W, r = self.comm.world, self.comm.rank
        dev = torch.device(self.device)
        user_emb, item_emb, ib, ub, gb = self.model.tables()      # full tables, in host memory (or wherever the model is)
        U, I = user_emb.shape[0], item_emb.shape[0]
        # every rank must hold the same initialisation (same seed -> same stream): checked, not assumed
        chk = torch.tensor([float(user_emb.double().sum()), float(item_emb.double().sum())], dtype=torch.float64, device=dev)
        lo_chk, hi_chk = chk.clone(), chk.clone()
        self.comm.all_reduce(lo_chk, op='max')
        hi_chk.neg_()
        self.comm.all_reduce(hi_chk, op='max')
        if not torch.equal(lo_chk, -hi_chk):
            raise RuntimeError('the ranks were built with different parameter initialisations (different seeds?)')
        lo, hi = item_range(I, r, W)
        to_dev = lambda t: t.contiguous().to(dev)                   # noqa: E731  (the shard only)
        user_emb, item_emb = to_dev(user_emb[r::W]), to_dev(item_emb[lo:hi])
        ib = None if ib is None else to_dev(ib[lo:hi])
        ub = None if ub is None else to_dev(ub[r::W])
        gb = None if gb is None else gb.to(dev)
        loader = self.train_loader
        arrays = loader.dataset.device_arrays(dev)
        # Batch semantics at N > 1.  conf['multi_gpu_batch']:
        #   'per_rank' (default)  every rank contributes train_batch_size positives, a step trains on world x
        #                         train_batch_size of them (weak scaling; lr unchanged -- NOT what the same conf does
        #                         on one GPU);
        #   'global'              a step trains on train_batch_size positives, train_batch_size / world per rank --
        #                         what nn.DataParallel does with the configured batch (train/trainer.py:38-41 of the
        #                         reference): the same optimisation as on one GPU.
        mode = conf.get('multi_gpu_batch', 'per_rank')
        if mode not in ('per_rank', 'global'):
            raise ValueError(f"multi_gpu_batch must be 'per_rank' or 'global', got {mode!r}")
        self._rank_batch = loader.batch_size if mode == 'per_rank' else max(1, loader.batch_size // self.comm.world)
        sh = ShardedBprMf(self.comm, user_emb, item_emb, ib, ub, gb, lr=self.lr, wd=self.wd,
                          batch=self._rank_batch, n_neg=loader.interaction_sampler.n_neg, seed=loader.seed,
                          loss=self.rec_loss.kind, log_adjust=getattr(self.rec_loss, 'log_adjust', 0.0),
                          alias=loader.interaction_sampler.alias(dev),
                          optimizer=conf['optimizer'], inputs_are_shards=True, n_users=U, n_items=I, **arrays)
        self._release_full_tables()
        return sh

    def _sharded_params(self):
        m = self.pointer_to_model
        out = [(m.user_embeddings.weight, 'user_emb', True), (m.item_embeddings.weight, 'item_emb', False)]
        if m.use_user_bias:
            out.append((m.user_bias.weight, 'user_bias', True))
        if m.use_item_bias:
            out.append((m.item_bias.weight, 'item_bias', False))
        return out

    def _release_full_tables(self):
        """The shards are the parameters now: the model's full tables are dropped (they come back, gathered from the
        shards, only while a checkpoint is written) so that no rank holds more than its share."""
        for p, _, _ in self._sharded_params():
            p.data = torch.empty((0,) + tuple(p.shape[1:]), dtype=p.dtype, device=p.device)

    def _sync_model_from_shards(self):
        """Assemble the sharded tables into the model's parameters (every rank), e.g. before saving."""
        full_u, full_ub = self.sharded.gather_user_table()
        full_i, full_ib = self.sharded.gather_item_table()
        full = {'user_emb': full_u, 'item_emb': full_i, 'user_bias': None if full_ub is None else full_ub.view(-1, 1),
                'item_bias': None if full_ib is None else full_ib.view(-1, 1)}
        for p, name, _ in self._sharded_params():
            p.data = full[name]
        m = self.pointer_to_model
        if m.use_global_bias:                       # replicated, gradient-free under BPR: every rank holds the same value
            m.global_bias.data = self.sharded.global_bias.detach().clone()

    def _save(self):
        if self.sharded is not None:
            self._sync_model_from_shards()
            if self.comm.rank == 0:
                self.pointer_to_model.save_model_to_path(self.model_path)
            self.comm.barrier()
            self._release_full_tables()
        else:
            self.pointer_to_model.save_model_to_path(self.model_path)

    def _train_epoch_sharded(self):
        """Global batch = world x train_batch_size positives per step; rank r takes the r-th slice.  The last
        (ragged) global batch is split evenly; at most world-1 interactions per epoch are left out."""
        loader, sh, W = self.train_loader, self.sharded, self.comm.world
        order = loader._epoch_order()
        if order is not None:
            self.comm.broadcast(order, src=0)   # one epoch order for the whole job
        loader.epoch += 1
        n, bs = len(loader.dataset), self._rank_batch
        steps, pos = 0, 0
        while n - pos >= W:
            nb = min(bs, (n - pos) // W)
            nxt = pos + nb * W                      # the loader knows its next batch: prepared a step ahead
            nnb = min(bs, (n - nxt) // W)
            sh.step_sampled(order, pos, nb, next_start=nxt if nnb > 0 else None, next_batch=nnb)
            pos = nxt
            steps += 1
            if steps == 1 or steps % 256 == 0:
                sh.check_status()                   # a capacity that is too small shows on the first batch; later
                                                    # overflows (the step is then invalid) within 256 steps
        sh.flush()
        rec = sh.pop_loss_sum() / max(steps, 1)
        sh.check_status()
        return {'epoch_train_loss': rec, 'epoch_train_rec_loss': rec, 'epoch_train_reg_loss': 0.0}

    def _log(self, log_dict):
        if self.use_wandb and not self._in_tune:
            wandb = _optional_module('wandb')
            if wandb is not None:
                wandb.log(log_dict)
        if self._in_tune:
            try:
                from ray.air import session
                session.report(log_dict)
            except ImportError:
                pass

    def _post_val(self, epoch, log_dict):
        hook = getattr(self.pointer_to_model, 'post_val', None)
        if callable(hook):
            log_dict.update(hook(epoch))

    # ------------------------------------------------------------------------------------------
    def _train_epoch_fused(self):
        loader, fused = self.train_loader, self.fused
        if isinstance(loader, TrainDataLoader):
            n_neg = loader.interaction_sampler.n_neg
            # the loader knows its next batch: each step samples + sorts it on the side stream (device-side
            # counterpart of the reference's DataLoader prefetch, train/trainer.py:127-130)
            batches = list(loader.fused_batches())
            k = 0
            while k < len(batches):
                order, start, nb = batches[k]
                run = 1                                   # consecutive full batches go down in one C call
                while (k + run < len(batches) and run < 256 and batches[k + run][2] == nb
                       and batches[k + run][1] == start + run * nb and batches[k + run][0] is order):
                    run += 1
                if k + run < len(batches) and run == 1:
                    fused.hint_next(*batches[k + 1], n_neg)
                if run > 1:
                    if k + run < len(batches):            # the run's last steps prepare the next run's first batches
                        nxt = batches[k + run]
                        two = (k + run + 1 < len(batches) and batches[k + run + 1][2] == nxt[2]
                               and batches[k + run + 1][1] == nxt[1] + nxt[2] and batches[k + run + 1][0] is nxt[0])
                        fused.hint_after_run(*nxt, n_neg, n_batches=2 if two else 1)
                    fused.steps_sampled(order, start, run, nb, n_neg)
                else:
                    fused.step_sampled(order, start, nb, n_neg)
                k += run
        else:
            for u_idxs, i_idxs, _labels in loader:
                fused.step(u_idxs.to(self.device), i_idxs.to(self.device))
        fused.flush()
        rec = fused.pop_loss_sum() / len(loader)
        fused.check_status()
        return {'epoch_train_loss': rec, 'epoch_train_rec_loss': rec, 'epoch_train_reg_loss': 0.0}

    def _train_epoch_autograd(self):
        sums = None
        for u_idxs, i_idxs, labels in self.train_loader:
            u_idxs, i_idxs = u_idxs.to(self.device), i_idxs.to(self.device)
            out = self.model(u_idxs, i_idxs)
            rec = self.rec_loss.compute_loss(out, labels)
            other = self.pointer_to_model.get_and_reset_other_loss()
            reg = other['reg_loss'].to(rec.device)
            total = rec + reg
            step = torch.stack([total.detach().reshape(()).double(), rec.detach().reshape(()).double(),
                                reg.detach().sum().double()])
            sums = step if sums is None else sums + step   # stays on the device: one sync per epoch
            total.backward()
            self.optimizer.step()
            self.optimizer.zero_grad()
        total, rec, reg = (sums / len(self.train_loader)).tolist()
        if hasattr(self.pointer_to_model, 'check_indices'):
            self.pointer_to_model.check_indices()
        return {'epoch_train_loss': total, 'epoch_train_rec_loss': rec, 'epoch_train_reg_loss': reg}

    def fit(self):
        patience = self.max_patience
        log_dict = self.val()
        self.best_value = log_dict['max_optimizing_metric'] = log_dict[self.optimizing_metric]
        self.best_epoch = log_dict['best_epoch'] = -1
        self.best_metrics = log_dict
        self._post_val(-1, log_dict)
        print('Init - Avg Val Value {:.3f} \n'.format(self.best_value))
        self._log(log_dict)
        self._save()

        for epoch in trange(self.n_epochs, disable=not self.batch_verbose):
            self.model.train()
            if patience == 0:
                print('Ran out of patience, Stopping ')
                break
            if self.sharded is not None:
                losses = self._train_epoch_sharded()
            elif self.fused is not None:
                losses = self._train_epoch_fused()
            else:
                losses = self._train_epoch_autograd()
            print('Epoch {} - Epoch Avg Train Loss {:.4f} ({:.4f} Rec Loss + {:.4f} Reg Loss )\n'.format(
                epoch, losses['epoch_train_loss'], losses['epoch_train_rec_loss'], losses['epoch_train_reg_loss']))

            metrics = self.val()
            current = metrics[self.optimizing_metric]
            print('Epoch {} - Avg Val Value {:.4f} \n'.format(epoch, current))
            if current > self.best_value:
                self.best_value = metrics['max_optimizing_metric'] = current
                self.best_epoch = metrics['best_epoch'] = epoch
                self.best_metrics = metrics
                print('Epoch {} - New best model found (val value {:.4f}) \n'.format(epoch, current))
                self._save()
                patience = self.max_patience
            else:
                metrics['max_optimizing_metric'] = self.best_value
                patience -= 1
            log_dict = {**metrics, **losses}
            self._post_val(epoch, log_dict)
            self._log(log_dict)
        if self.sharded is not None:
            self._sync_model_from_shards()   # the model leaves fit() with full tables, as on one GPU
        return self.best_metrics

    @torch.no_grad()
    def val(self):
        self.model.eval()
        if self.fused is not None:
            self.fused.flush()
        dataset = self.val_loader.dataset
        if self.sharded is not None:
            from hassaku_amd.dist import evaluate_item_sharded
            evaluator = FullEvaluator(aggr_by_group=True, n_groups=dataset.n_user_groups,
                                      user_to_user_group=dataset.user_to_user_group)
            return evaluate_item_sharded(self.comm, self.sharded, dataset, evaluator)
        evaluator = FullEvaluator(aggr_by_group=True, n_groups=dataset.n_user_groups,
                                  user_to_user_group=dataset.user_to_user_group)
        return evaluate_recommender_algorithm(self.pointer_to_model, self.val_loader, evaluator, self.device,
                                              self.batch_verbose)
