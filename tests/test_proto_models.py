"""ACF / UProtoMF / IProtoMF / UIProtoMF against golden vectors G10 written by the reference's
own classes (oracle/gen_golden.py::gen_g10): initialisation under the same seed, logits, extra losses, every parameter
gradient of bpr + reg_loss, parameters after two AdamW steps, evaluation-form scores.  The embedding gathers and their
dense backward run in libhassaku_hip.so (hsk_embedding_gather / hsk_embedding_backward), the loss in hsk_rec_loss_grad,
the optimiser in hsk_opt_dense."""
import numpy as np
import pytest
import torch

from conftest import assert_adam_param_close, load_golden, max_norm_err

MODELS = ['acf', 'uprotomf', 'iprotomf', 'uiprotomf']
RTOL = 1e-5


def _build(name):
    from hassaku_amd.algorithms.proto_alg import ACF, IProtoMF, UIProtoMF, UProtoMF
    U, I, D = 30, 80, 24
    return {'acf': lambda: ACF(U, I, D, 6, 0.1, 0.01),
            'uprotomf': lambda: UProtoMF(U, I, D, 7, 0.8, 0.6),
            'iprotomf': lambda: IProtoMF(U, I, D, 7, 0.8, 0.6),
            'uiprotomf': lambda: UIProtoMF(U, I, D, 5, 7, 0.8, 0.6, 0.7, 0.5)}[name]()


@pytest.mark.parametrize('name', MODELS)
def test_state_dict_and_seeded_init_match_reference(name):
    fx = load_golden(f'g10_{name}.npz')
    torch.manual_seed(64)
    sd = _build(name).state_dict()
    ref = {k[5:]: v for k, v in fx.items() if k.startswith('init.')}
    assert sorted(sd) == sorted(ref)
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k].shape, k
        assert np.array_equal(v.numpy(), ref[k]), f'{k}: initialisation differs from the reference under the same seed'


def test_registry_names():
    from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
    assert [m.name for m in AlgorithmsEnum] == ['mf', 'sgdbias', 'uprotomf', 'iprotomf', 'uiprotomf', 'acf']


@pytest.mark.gpu
@pytest.mark.parametrize('name', MODELS)
def test_two_training_steps_vs_golden(name):
    from hassaku_amd.train.optim import HipOptimizer
    from hassaku_amd.train.rec_losses import RecBayesianPersonalizedRankingLoss
    fx = load_golden(f'g10_{name}.npz')
    model = _build(name)
    model.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith('init.')})
    model = model.to('cuda')
    loss_fn = RecBayesianPersonalizedRankingLoss()
    opt = HipOptimizer(model.parameters(), 'adamw', lr=float(fx['lr']), weight_decay=float(fx['wd']))
    for step in (1, 2):
        u, i = torch.from_numpy(fx[f's{step}.u_idx']).cuda(), torch.from_numpy(fx[f's{step}.i_idx']).cuda()
        labels = torch.zeros(i.shape, dtype=torch.float64, device='cuda')
        labels[:, 0] = 1.
        out = model(u, i)
        ref = fx[f's{step}.logits']
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=RTOL, atol=RTOL * np.abs(ref).max())
        rec = loss_fn.compute_loss(out, labels)
        assert abs(rec.item() - float(fx[f's{step}.rec_loss'])) <= 1e-6 * abs(float(fx[f's{step}.rec_loss']))
        other = model.get_and_reset_other_loss()
        ref_other = {k.split('.other.')[1]: float(v) for k, v in fx.items() if k.startswith(f's{step}.other.')}
        assert sorted(other) == sorted(ref_other)
        for k, v in other.items():
            assert abs(float(v) - ref_other[k]) <= 1e-5 * max(abs(ref_other[k]), 1e-3), (k, float(v), ref_other[k])
        (rec + other['reg_loss']).backward()
        if step == 1:
            for pname, p in model.named_parameters():
                g = fx['s1.grad.' + pname]
                assert max_norm_err(p.grad.cpu().numpy(), g) < 2e-5, pname
        opt.step()
        opt.zero_grad()
        for k, v in model.state_dict().items():
            # tables of ~2000 elements: one element is 0.05 % -- the share beyond 1e-5 is allowed 1 % here
            assert_adam_param_close(v.cpu().numpy(), fx[f's{step}.param.' + k], f'{name} step {step} {k}', frac=0.01)
    model.check_indices()
    # the golden eval scores were taken after the two steps: compare against a model holding the step-2 parameters
    model.load_state_dict({k[9:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith('s2.param.')})
    with torch.no_grad():
        scores = model.combine_user_item_representations(
            model.get_user_representations(torch.from_numpy(fx['eval.u']).cuda()),
            model.get_item_representations(torch.arange(int(fx['n_items']), device='cuda')))
    ref = fx['eval.scores']
    np.testing.assert_allclose(scores.cpu().numpy(), ref, rtol=RTOL, atol=2e-6 * np.abs(ref).max())


@pytest.mark.gpu
def test_embedding_backward_is_the_dense_scatter_add():
    """hsk_embedding_backward == index_add in float64 (duplicates, untouched rows, 2-D index shapes, D % 4 != 0)."""
    from hassaku_amd import hip_ops
    g = torch.Generator(device='cuda').manual_seed(0)
    for n_rows, dim, shape in ((50, 24, (12, 6)), (1000, 402, (300,)), (7, 5, (4, 3, 2)), (3000, 64, (9000,))):
        table = torch.randn(n_rows, dim, device='cuda', generator=g, requires_grad=True)
        idx = torch.randint(0, n_rows, shape, device='cuda', generator=g)
        idx.view(-1)[: min(5, idx.numel())] = 3            # duplicates
        out = hip_ops.embedding(table, idx)
        assert torch.equal(out, table.detach()[idx])
        w = torch.randn(out.shape, device='cuda', generator=g)
        (out * w).sum().backward()
        ref = torch.zeros(n_rows, dim, dtype=torch.float64, device='cuda').index_add_(0, idx.view(-1), w.view(-1, dim).double())
        assert torch.allclose(table.grad.double(), ref, rtol=1e-6, atol=1e-6)
        assert (table.grad[ref.abs().sum(1) == 0] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize('alg_name, extra', [
    ('uprotomf', {'n_prototypes': 8, 'sim_proto_weight': 0.5, 'sim_batch_weight': 0.5}),
    ('acf', {'n_anchors': 6, 'delta_exc': 0.1, 'delta_inc': 0.01}),
])
def test_run_train_val_through_the_plugin_surface(tmp_path, alg_name, extra):
    """run_experiment's path for a prototype model: conf -> AlgorithmsEnum slot -> Trainer (autograd path: HIP gather,
    HIP loss, HIP optimiser) -> full evaluation (item representations once, HIP top-k + metrics) -> model.pth."""
    import os
    from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
    from hassaku_amd.data.data_utils import DatasetsEnum
    from hassaku_amd.data.synthetic import generate, write_csv_dataset
    from hassaku_amd.experiment_helper import run_train_val
    write_csv_dataset(generate(120, 250, 5000, seed=4, n_groups=2), str(tmp_path / 'data' / 'ml100k' / 'processed_dataset'))
    conf = {'data_path': str(tmp_path / 'data'), 'model_save_path': str(tmp_path / 'models'), 'embedding_dim': 16,
            'lr': 5e-3, 'wd': 1e-5, 'optimizer': 'adamw', 'n_epochs': 3, 'max_patience': 2, 'train_batch_size': 64,
            'neg_train': 6, 'rec_loss': 'bpr', 'eval_batch_size': 32, 'device': 'cuda',
            'running_settings': {'use_wandb': False, 'batch_verbose': False}, **extra}
    best, conf = run_train_val(AlgorithmsEnum[alg_name], DatasetsEnum.ml100k, conf)
    assert best['best_epoch'] >= -1 and np.isfinite(best['ndcg@10']) and 0 <= best['ndcg@10'] <= 1
    assert any(k.startswith('group_1_') for k in best)
    assert os.path.isfile(os.path.join(conf['model_path'], 'model.pth'))
