"""GPU end-to-end: the reference's plugin surface (model / loss / Trainer / evaluator / experiment helpers)
running on the HIP path."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_adam_param_close, csr_from_pairs, load_golden

pytestmark = pytest.mark.gpu


def _write_dataset(path, fx, groups=False):
    from hassaku_amd.data.synthetic import SyntheticInteractions, write_csv_dataset
    data = SyntheticInteractions(int(fx['n_users']), int(fx['n_items']), fx['train'], fx['val'], fx['test'],
                                 fx['user_group'] if groups else None)
    write_csv_dataset(data, path)
    return data


def _conf(model_path, **over):
    conf = {'device': 'cuda', 'lr': 3e-3, 'wd': 4e-5, 'optimizer': 'adamw', 'n_epochs': 2, 'optimizing_metric': 'ndcg@10',
            'max_patience': 1, 'model_path': model_path, 'train_batch_size': 64, 'neg_train': 6,
            'running_settings': {'use_wandb': False, 'batch_verbose': False, 'seed': 64}}
    conf.update(over)
    return conf


class _ReplayLoader:
    """A train loader that replays fixed (u, i, labels) batches, epoch after epoch."""

    def __init__(self, epochs):
        self.epochs, self.cur = epochs, 0
        self.dataset = None

    def __len__(self):
        return len(self.epochs[0])

    def __iter__(self):
        batches = self.epochs[min(self.cur, len(self.epochs) - 1)]
        self.cur += 1
        for u, i in batches:
            lab = torch.zeros(i.shape, dtype=torch.float64)
            lab[:, 0] = 1
            yield torch.from_numpy(u), torch.from_numpy(i), lab


def _g4_model_and_loaders(tmp_path, fx):
    from hassaku_amd.algorithms.sgd_alg import SGDMatrixFactorization
    from hassaku_amd.data.data_utils import get_dataloader
    _write_dataset(str(tmp_path), fx)
    model = SGDMatrixFactorization(int(fx['n_users']), int(fx['n_items']), int(fx['dim']), False, True, False)
    with torch.no_grad():
        for k in ('user_embeddings.weight', 'item_embeddings.weight', 'item_bias.weight'):
            model.state_dict()[k].copy_(torch.from_numpy(fx['init.' + k]))
    spe = int(fx['steps_per_epoch'])
    stream = [(fx[f'b{s}.u'], fx[f'b{s}.i']) for s in range(int(fx['n_steps']))]
    epochs = [stream[e * spe:(e + 1) * spe] for e in range(2)]
    val_loader = get_dataloader({'dataset_path': str(tmp_path), 'eval_batch_size': 16,
                                 'running_settings': {'eval_n_workers': 0}}, 'val')
    return model, _ReplayLoader(epochs), val_loader


@pytest.mark.parametrize('fused', [True, False])
def test_trainer_fit_replays_reference_run(tmp_path, fused):
    """G4 through Trainer.fit(): same batches as the reference run -> same parameters, same validation metrics,
    same best epoch; both for the fused step and for the autograd path (HIP fwd/bwd + torch.optim.AdamW)."""
    from hassaku_amd.train.rec_losses import RecBayesianPersonalizedRankingLoss
    from hassaku_amd.train.trainer import Trainer
    fx = load_golden('g4_fit.npz')
    model, train_loader, val_loader = _g4_model_and_loaders(tmp_path, fx)
    conf = _conf(str(tmp_path), lr=float(fx['lr']), wd=float(fx['wd']), fused_step=fused)
    trainer = Trainer(model, train_loader, val_loader, RecBayesianPersonalizedRankingLoss(), conf)
    assert (trainer.fused is not None) == fused
    best = trainer.fit()
    sd = {k: v.cpu().numpy() for k, v in model.state_dict().items()}
    for k in sd:
        assert_adam_param_close(sd[k], fx['final.' + k], k)
    assert best['best_epoch'] == int(fx['best_epoch'])
    names = [str(x) for x in fx['val_metric_names']]
    last = trainer.val()
    for name, val in zip(names, fx['val_metric_values_last']):
        assert abs(last[name] - val) <= 1e-6 + 2e-3 * abs(val), name
    assert abs(best['ndcg@10'] - fx['val_ndcg10'].max()) < 1e-4
    # model.pth written by save_model_to_path loads back with the reference's keys
    loaded = torch.load(os.path.join(str(tmp_path), 'model.pth'), map_location='cpu')
    assert sorted(loaded) == ['item_bias.weight', 'item_embeddings.weight', 'user_embeddings.weight']
    assert tuple(loaded['item_bias.weight'].shape) == (int(fx['n_items']), 1)


def test_model_forward_autograd_matches_golden():
    from hassaku_amd.algorithms.sgd_alg import SGDMatrixFactorization
    from hassaku_amd.train.rec_losses import RecBayesianPersonalizedRankingLoss
    fx = load_golden('g1_step_d64_all.npz')
    model = SGDMatrixFactorization(int(fx['n_users']), int(fx['n_items']), 64, True, True, True)
    model.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith('init.')})
    model.to('cuda')
    u, i = torch.from_numpy(fx['s1.u_idx']).cuda(), torch.from_numpy(fx['s1.i_idx']).cuda()
    out = model(u, i)
    np.testing.assert_allclose(out.detach().cpu().numpy(), fx['s1.logits'], rtol=1e-5, atol=1e-9)
    labels = torch.zeros(i.shape, dtype=torch.float64, device='cuda')
    labels[:, 0] = 1
    loss = RecBayesianPersonalizedRankingLoss().compute_loss(out, labels)
    assert loss.dtype == torch.float64 and abs(loss.item() - float(fx['s1.loss'])) < 1e-6
    (loss + model.get_and_reset_other_loss()['reg_loss'].to(loss.device)).backward()
    for name, p in model.named_parameters():
        ref = fx['s1.grad.' + name]
        assert p.grad is not None and tuple(p.grad.shape) == ref.shape and not p.grad.is_sparse
        if name in ('user_bias.weight', 'global_bias'):
            assert p.grad.abs().max().item() < 1e-5 * np.abs(fx['s1.grad_logits']).max()
        else:
            err = np.abs(p.grad.cpu().numpy() - ref).max() / np.abs(ref).max()
            assert err < 1e-5, (name, err)
    model.check_indices()
    # predict() and the evaluation form of combine_user_item_representations
    with torch.no_grad():
        i_repr = model.get_item_representations(torch.arange(model.n_items, device='cuda'))
        full = model.combine_user_item_representations(model.get_user_representations(u), i_repr)
    assert tuple(full.shape) == (len(u), model.n_items)
    picked = torch.gather(full, 1, i)
    np.testing.assert_allclose(picked.cpu().numpy(), model.predict(u, i).cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_device_loader_yields_reference_batch_contract(tmp_path):
    from hassaku_amd.data.dataloader import NegativeSampler, TrainDataLoader
    from hassaku_amd.data.dataset import TrainRecDataset
    fx = load_golden('g4_fit.npz')
    _write_dataset(str(tmp_path), fx)
    ds = TrainRecDataset(str(tmp_path))
    loader = TrainDataLoader(NegativeSampler(ds, 6), ds, batch_size=64, shuffle=True, device='cuda', seed=5)
    ptr, idx = csr_from_pairs(fx['train'], ds.n_users)
    seen = 0
    rows = []
    for u, i, lab in loader:
        assert u.dtype == torch.int64 and i.dtype == torch.int64 and lab.dtype == torch.float64
        assert i.shape == (len(u), 7) and lab.shape == i.shape and lab[:, 0].eq(1).all() and lab[:, 1:].eq(0).all()
        un, inn = u.cpu().numpy(), i.cpu().numpy()
        for b in range(len(un)):
            row = idx[ptr[un[b]]:ptr[un[b] + 1]]
            assert inn[b, 0] in row and not np.isin(inn[b, 1:], row).any()
        seen += len(u)
        rows.append(np.stack([un, inn[:, 0]], 1))
    assert seen == len(ds) and len(loader) == -(-len(ds) // 64)
    got = np.concatenate(rows)
    assert sorted(map(tuple, got)) == sorted(map(tuple, fx['train']))   # one epoch = every interaction once
    assert not np.array_equal(got, fx['train'])                          # ... in shuffled order


def test_run_train_val_test_end_to_end(tmp_path):
    """`run_experiment.py -a mf -d ml100k` equivalent on a synthetic dataset: learns (val ndcg improves over
    the untrained model), writes model.pth + conf.yml, test split evaluated from the saved model."""
    from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
    from hassaku_amd.data.data_utils import DatasetsEnum
    from hassaku_amd.data.synthetic import generate, write_csv_dataset
    from hassaku_amd.experiment_helper import run_train_val_test
    ds_path = str(tmp_path / 'data' / 'ml100k' / 'processed_dataset')
    write_csv_dataset(generate(300, 400, 12000, seed=2, n_groups=2), ds_path)
    conf = {'data_path': str(tmp_path / 'data'), 'model_save_path': str(tmp_path / 'models'), 'embedding_dim': 64,
            'lr': 5e-3, 'wd': 1e-5, 'use_user_bias': False, 'use_item_bias': True, 'use_global_bias': False,
            'optimizer': 'adamw', 'n_epochs': 6, 'max_patience': 5, 'train_batch_size': 128, 'neg_train': 10,
            'rec_loss': 'bpr', 'eval_batch_size': 256, 'device': 'cuda',
            'running_settings': {'use_wandb': False, 'train_n_workers': 0, 'batch_verbose': False}}
    best, test, conf = run_train_val_test(AlgorithmsEnum.mf, DatasetsEnum.ml100k, conf)
    assert best['best_epoch'] >= 0 and best['max_optimizing_metric'] == best['ndcg@10']
    assert best['ndcg@10'] > 0.02
    assert os.path.isfile(os.path.join(conf['model_path'], 'model.pth'))
    assert os.path.isfile(os.path.join(conf['model_path'], 'conf.yml'))
    assert len(test) == 36 and 'group_1_ndcg@100' in test and 0 < test['ndcg@10'] <= 1
    for k in ('epoch_train_loss', 'epoch_train_rec_loss', 'epoch_train_reg_loss'):
        pass  # per-epoch keys are printed/logged by Trainer.fit; best_metrics holds the validation dict


@pytest.mark.parametrize('opt,lr', [('adam', 5e-3), ('adagrad', 5e-2)])
def test_conf_optimizer_adam_adagrad_train_on_the_fused_step(tmp_path, opt, lr):
    """conf['optimizer'] = adam | adagrad (train/trainer.py:48-51): same driver, fused HIP step, and it learns."""
    from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
    from hassaku_amd.data.data_utils import DatasetsEnum
    from hassaku_amd.data.synthetic import generate, write_csv_dataset
    from hassaku_amd.experiment_helper import run_train_val_test
    from hassaku_amd.train import trainer as trainer_mod
    ds_path = str(tmp_path / 'data' / 'ml100k' / 'processed_dataset')
    write_csv_dataset(generate(300, 400, 12000, seed=2, n_groups=0), ds_path)
    conf = {'data_path': str(tmp_path / 'data'), 'model_save_path': str(tmp_path / 'models'), 'embedding_dim': 64,
            'lr': lr, 'wd': 1e-5, 'use_user_bias': False, 'use_item_bias': True, 'use_global_bias': False,
            'optimizer': opt, 'n_epochs': 6, 'max_patience': 5, 'train_batch_size': 128, 'neg_train': 10,
            'rec_loss': 'bpr', 'eval_batch_size': 256, 'device': 'cuda',
            'running_settings': {'use_wandb': False, 'train_n_workers': 0, 'batch_verbose': False}}
    built = []
    orig = trainer_mod.Trainer._build_fused

    def spy(self, c):
        built.append(c['optimizer'])
        return orig(self, c)

    trainer_mod.Trainer._build_fused = spy
    try:
        best, test, conf = run_train_val_test(AlgorithmsEnum.mf, DatasetsEnum.ml100k, conf)
    finally:
        trainer_mod.Trainer._build_fused = orig
    assert built == [opt]                      # the fused state was built for this optimiser (no autograd fallback)
    assert best['ndcg@10'] > 0.02 and 0 < test['ndcg@10'] <= 1
