"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the reference itself.

Runs ONLY in the authoring container (needs /root/reference); the fixtures are committed, the
reference never travels.  The reference is imported unmodified; modules it needs that are absent
offline and carry no arithmetic (wandb, ray, gdown) are replaced by empty in-process stand-ins, and
`T_co` (renamed in torch 2.10) is re-added to torch.utils.data.dataloader.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Fixtures (data only: inputs and the reference's outputs):
  g1_step_<tag>.npz   3 optimisation steps of SGDMatrixFactorization + RecBayesianPersonalizedRankingLoss +
                      torch.optim.AdamW on fixed (u,i) batches: logits, loss (fp64), dense grads of step 1,
                      params / exp_avg / exp_avg_sq after steps 1 and 3
  g3_eval.npz         evaluate_recommender_algorithm + FullEvaluator on a toy dataset with 2 user groups:
                      masked scores, top-100 ids, metric dict
  g4_fit.npz          reference Trainer.fit for 2 epochs with the reference loader (seed 64, 0 workers):
                      the batch stream, per-epoch losses, final parameters, validation metrics
  g5_metrics.npz      precision/recall/ndcg_at_k_batch on random logits/labels
  g8_opt_<tag>.npz    G1's protocol with torch.optim.Adam / torch.optim.Adagrad (conf['optimizer'] = adam | adagrad,
                      train/trainer.py:48-51): params after steps 1 and 3, exp_avg / exp_avg_sq or state_sum
  g9_sgdbias.npz      SGDBaseline (algorithms/sgd_alg.py:72-107) + RecBinaryCrossEntropy + AdamW, 3 steps: logits, loss,
                      dense grads of step 1, parameters after steps 1 and 3
  g7_checkpoint/      a checkpoint directory as the reference leaves it on disk: model.pth written by
                      save_model_to_path, conf.yml written by save_yaml after parse_conf, and expected.npz =
                      the saved tensors + the reference model's logits on a fixed (u, i) batch after
                      load_model_from_path into a fresh instance
"""
import os
import sys
import tempfile
import types
import typing

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')


def import_reference():
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    for name in ('wandb', 'ray', 'ray.air', 'ray.air.session', 'gdown'):
        sys.modules[name] = types.ModuleType(name)
    w = sys.modules['wandb']
    w.log = w.init = w.finish = lambda *a, **k: None
    sys.modules['ray'].air = sys.modules['ray.air']
    sys.modules['ray.air'].session = sys.modules['ray.air.session']
    sys.modules['ray.air.session'].report = lambda *a, **k: None
    import torch.utils.data.dataloader as dl
    dl.T_co = typing.TypeVar('T_co', covariant=True)


class _ExcludeAdapter:
    """scipy 1.15 dropped `.A` and torch-tensor indexing of CSR; eval/eval.py:250 uses both."""

    def __init__(self, csr):
        self.csr = csr

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.numpy()
        return types.SimpleNamespace(A=self.csr[idx].toarray())


def state_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def opt_state_np(model, opt):
    out = {}
    for name, p in model.named_parameters():
        st = opt.state[p]
        out['m.' + name] = st['exp_avg'].detach().numpy().copy()
        out['v.' + name] = st['exp_avg_sq'].detach().numpy().copy()
    return out


def gen_g1():
    from algorithms.sgd_alg import SGDMatrixFactorization
    from train.rec_losses import RecBayesianPersonalizedRankingLoss
    cases = [
        ('d16_item', 16, 40, 130, 12, 5, False, True, False),
        ('d64_item', 64, 40, 130, 16, 7, False, True, False),
        ('d402_item', 402, 24, 110, 8, 4, False, True, False),
        ('d64_all', 64, 40, 130, 16, 7, True, True, True),
        ('d30_none', 30, 33, 101, 9, 3, False, False, False),
        ('d64_dups', 64, 6, 20, 32, 9, False, True, False),   # heavy duplicate users / items in a batch
        ('d512_n100', 512, 16, 120, 8, 100, False, True, False),   # the BASELINE configs[2] row shape (D=512, 100 negatives)
    ]
    lr, wd = 3e-4, 4e-5
    for tag, D, U, I, B, N, ub, ib, gb in cases:
        torch.manual_seed(64)
        model = SGDMatrixFactorization(U, I, D, ub, ib, gb)
        loss_fn = RecBayesianPersonalizedRankingLoss()
        opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
        rng = np.random.RandomState(7)
        fx = {'lr': lr, 'wd': wd, 'n_users': U, 'n_items': I, 'dim': D,
              'use_user_bias': ub, 'use_item_bias': ib, 'use_global_bias': gb}
        for k, v in state_np(model).items():
            fx['init.' + k] = v
        for step in range(1, 4):
            u = torch.from_numpy(rng.randint(0, U, size=B).astype(np.int64))
            i = torch.from_numpy(rng.randint(0, I, size=(B, 1 + N)).astype(np.int64))
            labels = torch.zeros((B, 1 + N), dtype=torch.float64)
            labels[:, 0] = 1.
            out = model(u, i)
            loss = loss_fn.compute_loss(out, labels)
            reg = model.get_and_reset_other_loss()['reg_loss']
            total = loss + reg
            out.retain_grad()
            total.backward()
            fx[f's{step}.u_idx'] = u.numpy()
            fx[f's{step}.i_idx'] = i.numpy()
            fx[f's{step}.logits'] = out.detach().numpy().copy()
            fx[f's{step}.loss'] = np.array(loss.item(), dtype=np.float64)
            assert loss.dtype == torch.float64
            if step == 1:
                fx['s1.grad_logits'] = out.grad.numpy().copy()
                for name, p in model.named_parameters():
                    assert not p.grad.is_sparse
                    fx['s1.grad.' + name] = p.grad.numpy().copy()
            opt.step()
            opt.zero_grad()
            if step in (1, 3):
                for k, v in state_np(model).items():
                    fx[f's{step}.param.' + k] = v
                for k, v in opt_state_np(model, opt).items():
                    fx[f's{step}.' + k] = v
        np.savez_compressed(os.path.join(OUT, f'g1_step_{tag}.npz'), **fx)
        print('g1', tag, 'loss3', float(fx['s3.loss']))


def gen_g6():
    """bce and sampled_softmax on the same scorer (train/rec_losses.py:27-53,91-139): 3 AdamW steps each."""
    from algorithms.sgd_alg import SGDMatrixFactorization
    from train.rec_losses import RecBinaryCrossEntropy, RecSampledSoftmaxLoss
    cases = [
        ('bce_d32_item', 'bce', 32, 40, 130, 16, 7, False, True, False),
        ('bce_d32_all', 'bce', 32, 40, 130, 16, 7, True, True, True),
        ('ssm_d32_item', 'sampled_softmax', 32, 40, 130, 16, 7, False, True, False),
        ('ssm_d402_all', 'sampled_softmax', 402, 24, 110, 8, 4, True, True, True),
    ]
    lr, wd = 3e-4, 4e-5
    for tag, kind, D, U, I, B, N, ub, ib, gb in cases:
        torch.manual_seed(64)
        model = SGDMatrixFactorization(U, I, D, ub, ib, gb)
        with torch.no_grad():   # larger logits than the 0.1/D init gives, so the softmax / sigmoid are not flat
            model.user_embeddings.weight.mul_(20. * D ** 0.5)
            model.item_embeddings.weight.mul_(20. * D ** 0.5)
        loss_fn = RecBinaryCrossEntropy() if kind == 'bce' else RecSampledSoftmaxLoss(I, 'uniform', N)
        opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
        rng = np.random.RandomState(11)
        fx = {'lr': lr, 'wd': wd, 'n_users': U, 'n_items': I, 'dim': D, 'n_neg': N,
              'use_user_bias': ub, 'use_item_bias': ib, 'use_global_bias': gb,
              'log_adjust': float(np.log(I / N)) if kind == 'sampled_softmax' else 0.0}
        for k, v in state_np(model).items():
            fx['init.' + k] = v
        for step in range(1, 4):
            u = torch.from_numpy(rng.randint(0, U, size=B).astype(np.int64))
            i = torch.from_numpy(rng.randint(0, I, size=(B, 1 + N)).astype(np.int64))
            labels = torch.zeros((B, 1 + N), dtype=torch.float64)
            labels[:, 0] = 1.
            out = model(u, i)
            out.retain_grad()
            fx[f's{step}.logits'] = out.detach().numpy().copy()    # before the loss touches them (ssm adds in place)
            loss = loss_fn.compute_loss(out, labels)
            total = loss + model.get_and_reset_other_loss()['reg_loss']
            total.backward()
            fx[f's{step}.u_idx'] = u.numpy()
            fx[f's{step}.i_idx'] = i.numpy()
            fx[f's{step}.loss'] = np.array(loss.item(), dtype=np.float64)
            if step == 1:
                fx['s1.grad_logits'] = out.grad.numpy().copy()
                for name, p in model.named_parameters():
                    fx['s1.grad.' + name] = p.grad.numpy().copy()
            opt.step()
            opt.zero_grad()
            if step in (1, 3):
                for k, v in state_np(model).items():
                    fx[f's{step}.param.' + k] = v
                for k, v in opt_state_np(model, opt).items():
                    fx[f's{step}.' + k] = v
        np.savez_compressed(os.path.join(OUT, f'g6_{tag}.npz'), **fx)
        print('g6', tag, 'loss1', float(fx['s1.loss']), 'loss3', float(fx['s3.loss']), loss.dtype)


def toy_dataset(tmp, n_users=64, n_items=150, n_inter=2600, n_groups=2, seed=3):
    from hassaku_amd.data.synthetic import generate, write_csv_dataset
    data = generate(n_users, n_items, n_inter, seed=seed, n_groups=n_groups)
    write_csv_dataset(data, tmp)
    return data


def gen_g3():
    from algorithms.sgd_alg import SGDMatrixFactorization
    from data.dataset import FullEvalDataset
    from eval.eval import evaluate_recommender_algorithm, FullEvaluator
    from torch.utils.data import DataLoader
    with tempfile.TemporaryDirectory() as tmp:
        data = toy_dataset(tmp)
        fx = {'n_users': data.n_users, 'n_items': data.n_items, 'train': data.train, 'val': data.val,
              'test': data.test, 'user_group': data.user_group}
        for split in ('val', 'test'):
            ds = FullEvalDataset(tmp, split)
            raw_excl = ds.exclude_data
            ds.exclude_data = _ExcludeAdapter(raw_excl)
            loader = DataLoader(ds, batch_size=16, num_workers=0)
            torch.manual_seed(64)
            model = SGDMatrixFactorization(data.n_users, data.n_items, 48, True, True, True)
            with torch.no_grad():  # spread the scores out (init std is 0.1/D)
                model.user_embeddings.weight.mul_(300.)
                model.item_embeddings.weight.mul_(300.)
                model.global_bias.fill_(0.25)
            captured = {}

            class Spy(FullEvaluator):
                def eval_batch(self, u_idxs, logits, y_true):
                    captured.setdefault('u', []).append(u_idxs.numpy().copy())
                    captured.setdefault('logits', []).append(logits.numpy().copy())
                    captured.setdefault('topk', []).append(logits.topk(100).indices.numpy().copy())
                    super().eval_batch(u_idxs, logits, y_true)

            ev = Spy(aggr_by_group=True, n_groups=ds.n_user_groups, user_to_user_group=ds.user_to_user_group)
            metrics = evaluate_recommender_algorithm(model, loader, ev, 'cpu', False)
            if split == 'val':
                for k, v in state_np(model).items():
                    fx['param.' + k] = v
            fx[f'{split}.u'] = np.concatenate(captured['u'])
            fx[f'{split}.masked_logits'] = np.concatenate(captured['logits'])
            fx[f'{split}.top100'] = np.concatenate(captured['topk'])
            fx[f'{split}.metric_names'] = np.array(sorted(metrics))
            fx[f'{split}.metric_values'] = np.array([metrics[k] for k in sorted(metrics)], dtype=np.float64)
            print('g3', split, 'ndcg@10', metrics['ndcg@10'], 'n metrics', len(metrics))
        np.savez_compressed(os.path.join(OUT, 'g3_eval.npz'), **fx)


def gen_g3_d512():
    """G3 at the BASELINE embedding size and a catalogue wide enough (> 4096 items) for the wide-row top-k paths:
    D = 512, 4224 items, 48 users.  The item table alone is 8.6 MB, so the fixture does not carry the parameters:
    they are the reference's own seeded initialisation (torch.manual_seed(64), scaled as below) -- which
    hassaku_amd reproduces bit for bit (tests/test_host_logic.py) -- and the fixture holds a checksum of them next to
    the reference's outputs."""
    from algorithms.sgd_alg import SGDMatrixFactorization
    from data.dataset import FullEvalDataset
    from eval.eval import evaluate_recommender_algorithm, FullEvaluator
    from torch.utils.data import DataLoader
    with tempfile.TemporaryDirectory() as tmp:
        data = toy_dataset(tmp, n_users=48, n_items=4224, n_inter=6000, n_groups=2, seed=9)
        ds = FullEvalDataset(tmp, 'val')
        ds.exclude_data = _ExcludeAdapter(ds.exclude_data)
        loader = DataLoader(ds, batch_size=16, num_workers=0)
        torch.manual_seed(64)
        model = SGDMatrixFactorization(data.n_users, data.n_items, 512, False, True, False)
        with torch.no_grad():  # spread the scores out (init std is 0.1/D)
            model.user_embeddings.weight.mul_(2000.)
            model.item_embeddings.weight.mul_(2000.)
        captured = {}

        class Spy(FullEvaluator):
            def eval_batch(self, u_idxs, logits, y_true):
                captured.setdefault('u', []).append(u_idxs.numpy().copy())
                captured.setdefault('logits', []).append(logits.numpy().copy())
                captured.setdefault('topk', []).append(logits.topk(100).indices.numpy().copy())
                super().eval_batch(u_idxs, logits, y_true)

        ev = Spy(aggr_by_group=True, n_groups=ds.n_user_groups, user_to_user_group=ds.user_to_user_group)
        metrics = evaluate_recommender_algorithm(model, loader, ev, 'cpu', False)
        sd = state_np(model)
        fx = {'n_users': data.n_users, 'n_items': data.n_items, 'dim': 512, 'seed': 64, 'scale': 2000.0,
              'train': data.train, 'val': data.val, 'user_group': data.user_group,
              'param_checksum': np.array([np.float64(v.astype(np.float64).sum()) for k, v in sorted(sd.items())]),
              'param_names': np.array(sorted(sd)),
              'item_row_17': sd['item_embeddings.weight'][17], 'user_row_5': sd['user_embeddings.weight'][5],
              'val.u': np.concatenate(captured['u']), 'val.masked_logits': np.concatenate(captured['logits']),
              'val.top100': np.concatenate(captured['topk']),
              'val.metric_names': np.array(sorted(metrics)),
              'val.metric_values': np.array([metrics[k] for k in sorted(metrics)], dtype=np.float64)}
        print('g3_d512 ndcg@10', metrics['ndcg@10'], 'n metrics', len(metrics))
        np.savez_compressed(os.path.join(OUT, 'g3_eval_d512.npz'), **fx)


def gen_g4():
    from algorithms.sgd_alg import SGDMatrixFactorization
    from data.dataloader import TrainDataLoader, NegativeSampler
    from data.dataset import TrainRecDataset, FullEvalDataset
    from train.rec_losses import RecBayesianPersonalizedRankingLoss
    from train.trainer import Trainer
    from utilities.utils import reproducible
    from torch.utils.data import DataLoader
    with tempfile.TemporaryDirectory() as tmp:
        data = toy_dataset(tmp, n_users=48, n_items=120, n_inter=1500, n_groups=0, seed=5)
        reproducible(64)
        train_ds = TrainRecDataset(tmp)
        n_neg, bs = 6, 64
        loader = TrainDataLoader(NegativeSampler(train_ds, n_neg, 'uniform'), train_ds, batch_size=bs, shuffle=True,
                                 num_workers=0, prefetch_factor=None)
        val_ds = FullEvalDataset(tmp, 'val')
        val_ds.exclude_data = _ExcludeAdapter(val_ds.exclude_data)
        val_loader = DataLoader(val_ds, batch_size=16, num_workers=0)
        model = SGDMatrixFactorization(train_ds.n_users, train_ds.n_items, 32, False, True, False)
        init = state_np(model)
        conf = {'device': 'cpu', 'lr': 3e-3, 'wd': 4e-5, 'optimizer': 'adamw', 'n_epochs': 2,
                'optimizing_metric': 'ndcg@10', 'max_patience': 1, 'model_path': tmp,
                'running_settings': {'use_wandb': False, 'batch_verbose': False}}
        stream = []

        class Recorder:
            """forwards the loader and records every batch it yields"""

            def __init__(self, inner):
                self.inner = inner
                self.dataset = inner.dataset

            def __len__(self):
                return len(self.inner)

            def __iter__(self):
                for u, i, lab in self.inner:
                    stream.append((u.numpy().copy(), i.numpy().copy()))
                    assert lab.dtype == torch.float64
                    yield u, i, lab

        trainer = Trainer(model, Recorder(loader), val_loader, RecBayesianPersonalizedRankingLoss(), conf)
        epoch_logs = []
        orig_val = trainer.val

        def val_spy():
            m = orig_val()
            epoch_logs.append(dict(m))
            return m

        trainer.val = val_spy
        best = trainer.fit()
        fx = {'n_users': data.n_users, 'n_items': data.n_items, 'train': data.train, 'val': data.val,
              'test': data.test, 'n_neg': n_neg, 'batch_size': bs, 'lr': conf['lr'], 'wd': conf['wd'], 'dim': 32,
              'n_steps': len(stream), 'steps_per_epoch': len(loader)}
        for k, v in init.items():
            fx['init.' + k] = v
        for k, v in state_np(model).items():
            fx['final.' + k] = v
        for s, (u, i) in enumerate(stream):
            fx[f'b{s}.u'] = u
            fx[f'b{s}.i'] = i
        fx['val_ndcg10'] = np.array([m['ndcg@10'] for m in epoch_logs], dtype=np.float64)
        fx['val_metric_names'] = np.array(sorted(epoch_logs[-1]))
        fx['val_metric_values_last'] = np.array([epoch_logs[-1][k] for k in sorted(epoch_logs[-1])], dtype=np.float64)
        fx['best_epoch'] = np.array(best['best_epoch'])
        np.savez_compressed(os.path.join(OUT, 'g4_fit.npz'), **fx)
        print('g4 steps', len(stream), 'val ndcg@10 per val call', fx['val_ndcg10'])


def gen_g5():
    from eval.metrics import precision_at_k_batch, recall_at_k_batch, ndcg_at_k_batch
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(12, 300, generator=g)
    y = (torch.rand(12, 300, generator=g) < 0.04).float()
    y[3] = 0.  # a user without ground truth
    y[5, :] = 0.
    y[5, 17] = 1.
    fx = {'logits': logits.numpy(), 'y_true': y.numpy()}
    for k in (5, 10, 50, 100):
        fx[f'precision@{k}'] = precision_at_k_batch(logits, y, k, aggr_sum=False).numpy()
        fx[f'recall@{k}'] = recall_at_k_batch(logits, y, k, aggr_sum=False).numpy()
        fx[f'ndcg@{k}'] = ndcg_at_k_batch(logits, y, k, aggr_sum=False).numpy()
    fx['top100'] = logits.topk(100).indices.numpy()
    np.savez_compressed(os.path.join(OUT, 'g5_metrics.npz'), **fx)
    print('g5 ok')


def gen_g8():
    from algorithms.sgd_alg import SGDMatrixFactorization
    from train.rec_losses import RecBayesianPersonalizedRankingLoss
    cases = [('adam_d32_item', 'adam', 32, 40, 130, 16, 7, False, True, False, 3e-4, 1e-3),
             ('adam_d64_all', 'adam', 64, 40, 130, 16, 7, True, True, True, 3e-4, 1e-3),
             ('adagrad_d32_item', 'adagrad', 32, 40, 130, 16, 7, False, True, False, 1e-2, 1e-3),
             ('adagrad_d402_all', 'adagrad', 402, 24, 110, 8, 4, True, True, True, 1e-2, 1e-3)]
    for tag, opt_name, D, U, I, B, N, ub, ib, gb, lr, wd in cases:
        torch.manual_seed(64)
        model = SGDMatrixFactorization(U, I, D, ub, ib, gb)
        loss_fn = RecBayesianPersonalizedRankingLoss()
        cls = {'adam': torch.optim.Adam, 'adagrad': torch.optim.Adagrad}[opt_name]
        opt = cls(model.parameters(), lr=lr, weight_decay=wd)             # exactly train/trainer.py:48-51
        rng = np.random.RandomState(11)
        fx = {'lr': lr, 'wd': wd, 'n_users': U, 'n_items': I, 'dim': D, 'optimizer': opt_name,
              'use_user_bias': ub, 'use_item_bias': ib, 'use_global_bias': gb}
        for k, v in state_np(model).items():
            fx['init.' + k] = v
        for step in range(1, 4):
            u = torch.from_numpy(rng.randint(0, U, size=B).astype(np.int64))
            i = torch.from_numpy(rng.randint(0, I, size=(B, 1 + N)).astype(np.int64))
            labels = torch.zeros((B, 1 + N), dtype=torch.float64)
            labels[:, 0] = 1.
            out = model(u, i)
            loss = loss_fn.compute_loss(out, labels)
            total = loss + model.get_and_reset_other_loss()['reg_loss']
            total.backward()
            fx[f's{step}.u_idx'] = u.numpy()
            fx[f's{step}.i_idx'] = i.numpy()
            fx[f's{step}.loss'] = np.array(loss.item(), dtype=np.float64)
            if step == 1:
                for name, p in model.named_parameters():
                    fx['s1.grad.' + name] = p.grad.numpy().copy()
            opt.step()
            opt.zero_grad()
            if step in (1, 3):
                for k, v in state_np(model).items():
                    fx[f's{step}.param.' + k] = v
                for name, p in model.named_parameters():
                    st = opt.state[p]
                    if opt_name == 'adagrad':
                        fx[f's{step}.v.' + name] = st['sum'].detach().numpy().copy()
                    else:
                        fx[f's{step}.m.' + name] = st['exp_avg'].detach().numpy().copy()
                        fx[f's{step}.v.' + name] = st['exp_avg_sq'].detach().numpy().copy()
        np.savez_compressed(os.path.join(OUT, f'g8_opt_{tag}.npz'), **fx)
        print('g8', tag, 'ok')


def gen_g9():
    from algorithms.sgd_alg import SGDBaseline
    from train.rec_losses import RecBinaryCrossEntropy
    U, I, B, N, lr, wd = 40, 130, 16, 7, 3e-3, 4e-5
    torch.manual_seed(64)
    model = SGDBaseline(U, I)
    loss_fn = RecBinaryCrossEntropy()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
    rng = np.random.RandomState(13)
    fx = {'lr': lr, 'wd': wd, 'n_users': U, 'n_items': I}
    for k, v in state_np(model).items():
        fx['init.' + k] = v
    for step in range(1, 4):
        u = torch.from_numpy(rng.randint(0, U, size=B).astype(np.int64))
        i = torch.from_numpy(rng.randint(0, I, size=(B, 1 + N)).astype(np.int64))
        labels = torch.zeros((B, 1 + N), dtype=torch.float64)
        labels[:, 0] = 1.
        out = model(u, i)
        loss = loss_fn.compute_loss(out, labels)
        total = loss + model.get_and_reset_other_loss()['reg_loss']
        total.backward()
        fx[f's{step}.u_idx'] = u.numpy()
        fx[f's{step}.i_idx'] = i.numpy()
        fx[f's{step}.logits'] = out.detach().numpy().copy()
        fx[f's{step}.loss'] = np.array(loss.item(), dtype=np.float64)
        if step == 1:
            for name, p in model.named_parameters():
                fx['s1.grad.' + name] = p.grad.numpy().copy()
        opt.step()
        opt.zero_grad()
        if step in (1, 3):
            for k, v in state_np(model).items():
                fx[f's{step}.param.' + k] = v
    np.savez_compressed(os.path.join(OUT, 'g9_sgdbias.npz'), **fx)
    print('g9 ok', sorted(k for k in fx if k.startswith('init.')))


def gen_g10():
    """G10: the anchor / prototype models (ACF, UProtoMF, IProtoMF, UIProtoMF -- SURVEY 8f rank 4): seeded
    initialisation, logits of a batch, the extra losses, all parameter gradients of (bpr + reg_loss), parameters after
    two AdamW steps, and the evaluation-form scores (every user against the whole item list)."""
    from algorithms.sgd_alg import ACF, UProtoMF, IProtoMF, UIProtoMF
    from train.rec_losses import RecBayesianPersonalizedRankingLoss
    U, I, D, B, N, lr, wd = 30, 80, 24, 12, 5, 1e-3, 1e-4
    builders = {
        'acf': lambda: ACF(U, I, D, 6, 0.1, 0.01),
        'uprotomf': lambda: UProtoMF(U, I, D, 7, 0.8, 0.6),
        'iprotomf': lambda: IProtoMF(U, I, D, 7, 0.8, 0.6),
        'uiprotomf': lambda: UIProtoMF(U, I, D, 5, 7, 0.8, 0.6, 0.7, 0.5),
    }
    for name, build in builders.items():
        torch.manual_seed(64)
        model = build()
        fx = {'n_users': U, 'n_items': I, 'dim': D, 'lr': lr, 'wd': wd}
        for k, v in state_np(model).items():
            fx['init.' + k] = v
        rng = np.random.RandomState(3)
        loss_fn = RecBayesianPersonalizedRankingLoss()
        opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
        for step in (1, 2):
            u = torch.from_numpy(rng.randint(0, U, size=B).astype(np.int64))
            i = torch.from_numpy(rng.randint(0, I, size=(B, N + 1)).astype(np.int64))
            labels = torch.zeros((B, N + 1), dtype=torch.float64)
            labels[:, 0] = 1.
            out = model(u, i)
            rec = loss_fn.compute_loss(out, labels)
            other = model.get_and_reset_other_loss()
            total = rec + other['reg_loss']
            fx[f's{step}.u_idx'], fx[f's{step}.i_idx'] = u.numpy(), i.numpy()
            fx[f's{step}.logits'] = out.detach().numpy().copy()
            fx[f's{step}.rec_loss'] = np.float64(rec.item())
            for k, v in other.items():
                fx[f's{step}.other.{k}'] = np.float64(float(v))
            total.backward()
            if step == 1:
                for pname, p in model.named_parameters():
                    fx['s1.grad.' + pname] = p.grad.numpy().copy()
            opt.step()
            opt.zero_grad()
            for k, v in state_np(model).items():
                fx[f's{step}.param.' + k] = v
        with torch.no_grad():   # evaluation form, eval/eval.py:237-248
            ue = torch.arange(0, 9)
            scores = model.combine_user_item_representations(model.get_user_representations(ue),
                                                             model.get_item_representations(torch.arange(I)))
            model.get_and_reset_other_loss()
        fx['eval.u'] = ue.numpy()
        fx['eval.scores'] = scores.numpy().copy()
        np.savez_compressed(os.path.join(OUT, f'g10_{name}.npz'), **fx)
        print('g10', name, 'rec', fx['s1.rec_loss'], 'reg', fx['s1.other.reg_loss'])


def gen_g7():
    from algorithms.algorithms_utils import AlgorithmsEnum
    from algorithms.sgd_alg import SGDMatrixFactorization
    from conf.conf_parser import parse_conf, save_yaml
    from data.data_utils import DatasetsEnum
    from utilities.utils import reproducible
    out = os.path.join(OUT, 'g7_checkpoint')
    os.makedirs(out, exist_ok=True)
    reproducible(7)
    conf = {'embedding_dim': 24, 'use_user_bias': True, 'use_item_bias': True, 'use_global_bias': True,
            'rec_loss': 'bpr', 'lr': 3e-4, 'wd': 4e-5, 'optimizer': 'adamw', 'neg_train': 10, 'train_batch_size': 32,
            'data_path': 'unused', 'running_settings': {'use_wandb': False}}
    conf = parse_conf(conf, AlgorithmsEnum.mf, DatasetsEnum.ml1m)
    conf['model_path'] = 'g7_checkpoint'            # the reference stores an absolute run directory here
    conf['dataset_path'] = 'unused'
    ds = types.SimpleNamespace(n_users=37, n_items=53)
    model = SGDMatrixFactorization.build_from_conf(conf, ds)
    with torch.no_grad():                            # trained-looking values everywhere, biases included
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.3)
    model.save_model_to_path(out)
    save_yaml(out, conf)
    fresh = SGDMatrixFactorization.build_from_conf(conf, ds)
    fresh.load_model_from_path(out)
    g = torch.Generator().manual_seed(3)
    u = torch.randint(0, 37, (19,), generator=g)
    i = torch.randint(0, 53, (19, 8), generator=g)
    fx = {'u_idx': u.numpy(), 'i_idx': i.numpy(), 'logits': fresh.predict(u, i).numpy()}
    fx.update({'sd.' + k: v for k, v in state_np(fresh).items()})
    np.savez_compressed(os.path.join(out, 'expected.npz'), **fx)
    print('g7 ok', sorted(k for k in fx if k.startswith('sd.')))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    import_reference()
    torch.set_num_threads(1)
    only = sys.argv[1:]          # e.g. `gen_golden.py g3_d512`: regenerate a single fixture
    if only:
        for name in only:
            globals()['gen_' + name]()
        sys.exit(0)
    gen_g1()
    gen_g3()
    gen_g3_d512()
    gen_g4()
    gen_g5()
    gen_g6()
    gen_g7()
    gen_g8()
    gen_g9()
    gen_g10()
