"""SGDBaseline -- `-a sgdbias` (algorithms/sgd_alg.py:72-107 of the reference): the bias-only
sibling of MF on the same plugin API: hsk_mf_scores / hsk_mf_backward with dim = 0, loss and optimiser kernels of the
autograd path.  Golden G9 = the reference's SGDBaseline + RecBinaryCrossEntropy + torch.optim.AdamW, three steps."""
import numpy as np
import pytest
import torch

from conftest import assert_adam_param_close, load_golden, max_norm_err

KEYS = {'user_bias.weight': 'user_bias', 'item_bias.weight': 'item_bias', 'global_bias': 'global_bias'}


def test_oracle_bias_model_matches_reference(oracle):
    fx = load_golden('g9_sgdbias.npz')
    tr = oracle.BiasOracleTrainer(fx['init.user_bias.weight'], fx['init.item_bias.weight'], fx['init.global_bias'],
                                  lr=float(fx['lr']), wd=float(fx['wd']), loss='bce')
    for step in (1, 2, 3):
        loss, logits, _, grads = tr.step(fx[f's{step}.u_idx'], fx[f's{step}.i_idx'])
        np.testing.assert_allclose(logits, fx[f's{step}.logits'], rtol=1e-5, atol=2e-6)
        assert abs(loss - float(fx[f's{step}.loss'])) <= 1e-6 * abs(float(fx[f's{step}.loss']))
        if step == 1:
            for sk, name in KEYS.items():
                assert max_norm_err(grads[name], fx['s1.grad.' + sk].reshape(-1)) < 1e-5, name
        if step in (1, 3):
            for sk, name in KEYS.items():
                assert_adam_param_close(tr.P[name], fx[f's{step}.param.{sk}'], (step, name))


def test_registry_and_state_dict_match_reference():
    from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
    from hassaku_amd.algorithms.sgd_alg import SGDBaseline
    assert AlgorithmsEnum['sgdbias'].value is SGDBaseline
    fx = load_golden('g9_sgdbias.npz')
    torch.manual_seed(64)
    m = SGDBaseline(int(fx['n_users']), int(fx['n_items']))
    sd = m.state_dict()
    assert list(sd.keys()) == ['global_bias', 'user_bias.weight', 'item_bias.weight']
    for k, v in sd.items():                              # same seed -> bit-identical initial parameters
        assert np.array_equal(v.numpy(), fx['init.' + k]), k
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, dtype=torch.int64), torch.zeros((2, 3), dtype=torch.int64))   # no CPU forward


@pytest.mark.gpu
def test_hip_sgdbias_three_steps_vs_golden():
    from hassaku_amd.algorithms.sgd_alg import SGDBaseline
    from hassaku_amd.train.optim import HipOptimizer
    from hassaku_amd.train.rec_losses import RecBinaryCrossEntropy
    fx = load_golden('g9_sgdbias.npz')
    model = SGDBaseline(int(fx['n_users']), int(fx['n_items']))
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(torch.from_numpy(fx['init.' + k]))
    model = model.to('cuda')
    loss_fn = RecBinaryCrossEntropy()
    opt = HipOptimizer(model.parameters(), 'adamw', lr=float(fx['lr']), weight_decay=float(fx['wd']))
    for step in (1, 2, 3):
        u, i = torch.from_numpy(fx[f's{step}.u_idx']).cuda(), torch.from_numpy(fx[f's{step}.i_idx']).cuda()
        labels = torch.zeros(i.shape, dtype=torch.float64, device='cuda')
        labels[:, 0] = 1.
        out = model(u, i)
        loss = loss_fn.compute_loss(out, labels)
        np.testing.assert_allclose(out.detach().cpu().numpy(), fx[f's{step}.logits'], rtol=1e-5, atol=2e-6)
        assert abs(loss.item() - float(fx[f's{step}.loss'])) <= 1e-6 * abs(float(fx[f's{step}.loss']))
        (loss + model.get_and_reset_other_loss()['reg_loss'].to(loss.device)).backward()
        if step == 1:
            for name, p in model.named_parameters():
                assert max_norm_err(p.grad.cpu().numpy(), fx['s1.grad.' + name]) < 1e-5, name
        opt.step()
        opt.zero_grad()
        if step in (1, 3):
            for k, v in model.state_dict().items():
                assert_adam_param_close(v.cpu().numpy(), fx[f's{step}.param.{k}'], (step, k))
    model.check_indices()


@pytest.mark.gpu
def test_sgdbias_runs_through_the_experiment_driver(tmp_path):
    """`run_experiment.py -a sgdbias` equivalent: trains on the autograd path with the HIP optimiser and is evaluated
    by the generic (non-MF) branch of evaluate_recommender_algorithm; a popularity-like ranking beats chance."""
    from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
    from hassaku_amd.data.data_utils import DatasetsEnum
    from hassaku_amd.data.synthetic import generate, write_csv_dataset
    from hassaku_amd.experiment_helper import run_train_val_test
    ds_path = str(tmp_path / 'data' / 'ml100k' / 'processed_dataset')
    write_csv_dataset(generate(200, 300, 9000, seed=4, n_groups=0), ds_path)
    conf = {'data_path': str(tmp_path / 'data'), 'model_save_path': str(tmp_path / 'models'), 'lr': 2e-2, 'wd': 1e-6,
            'optimizer': 'adamw', 'n_epochs': 4, 'max_patience': 3, 'train_batch_size': 256, 'neg_train': 8,
            'rec_loss': 'bce', 'eval_batch_size': 128, 'device': 'cuda',
            'running_settings': {'use_wandb': False, 'train_n_workers': 0, 'batch_verbose': False}}
    best, test, conf = run_train_val_test(AlgorithmsEnum.sgdbias, DatasetsEnum.ml100k, conf)
    assert best['ndcg@10'] > 0.01 and 0 < test['ndcg@10'] <= 1
