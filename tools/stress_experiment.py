#!/usr/bin/env python3
"""Randomised end-to-end runs through the reference's own entry point shape (experiment_helper.run_train_val_test: conf ->
dataset -> model -> Trainer.fit -> evaluation -> model.pth / conf.yml -> test split) on small synthetic datasets:
    python tools/stress_experiment.py [seconds] [seed]          (on a GPU box)
Random confs inside the reference's vocabulary (embedding_dim of every alignment, batch sizes from 1 to beyond the number of
interactions, 1 .. 150 negatives, bpr / bce / sampled_softmax, adamw / adam / adagrad, uniform / popular negatives, bias
switches, eval batch sizes, user groups or none).  A run passes when it raises nothing, its metrics are finite numbers in
[0, 1], the training loss is finite and the saved model loads back to the same validation metrics."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one_case(rng, root):
    from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
    from hassaku_amd.data.data_utils import DatasetsEnum
    from hassaku_amd.data.synthetic import generate, write_csv_dataset
    from hassaku_amd.experiment_helper import run_train_val_test
    n_users, n_items = int(rng.randint(30, 500)), int(rng.randint(160, 900))
    nnz = int(rng.randint(12 * n_users, 40 * n_users))
    groups = int(rng.choice([0, 2, 3]))
    work = tempfile.mkdtemp(dir=root)
    ds_path = os.path.join(work, 'data', 'ml100k', 'processed_dataset')
    write_csv_dataset(generate(n_users, n_items, nnz, seed=int(rng.randint(1 << 30)), n_groups=groups), ds_path)
    loss = str(rng.choice(['bpr', 'bpr', 'bce', 'sampled_softmax']))
    conf = {'data_path': os.path.join(work, 'data'), 'model_save_path': os.path.join(work, 'models'),
            'embedding_dim': int(rng.choice([1, 2, 7, 16, 30, 64, 100, 128, 402])),
            'lr': float(10 ** rng.uniform(-3.5, -2)), 'wd': float(rng.choice([0.0, 1e-5, 1e-3])),
            'use_user_bias': bool(rng.rand() < 0.3) and loss != 'bce', 'use_item_bias': bool(rng.rand() < 0.7),
            'use_global_bias': bool(rng.rand() < 0.3) and loss != 'bce',
            'optimizer': str(rng.choice(['adamw', 'adamw', 'adam', 'adagrad'])), 'n_epochs': 0, 'max_patience': 0, 'train_batch_size': int(rng.choice([1, 7, 64, 128, 1000, 5000, 100000])),
            'neg_train': int(rng.choice([1, 2, 10, 50, 150])), 'rec_loss': loss,
            'train_neg_strategy': str(rng.choice(['uniform', 'uniform', 'popular'])),
            'eval_batch_size': int(rng.choice([1, 16, 256, 100000])), 'device': 'cuda',
            'running_settings': {'use_wandb': False, 'train_n_workers': 0, 'batch_verbose': False,
                                 'seed': int(rng.randint(1 << 20))}}
    conf['n_epochs'] = int(rng.randint(2, 5))      # (conf_parser, as the reference: 0 < max_patience < n_epochs)
    conf['max_patience'] = int(rng.randint(1, conf['n_epochs']))
    if conf['train_batch_size'] == 1 and nnz > 3000:
        conf['train_batch_size'] = 3               # (one positive per step: keep the run in seconds)
    alg = str(rng.choice(['mf', 'mf', 'mf', 'sgdbias', 'uprotomf', 'iprotomf', 'uiprotomf', 'acf']))
    conf.update(n_prototypes=int(rng.choice([2, 5, 20])), sim_proto_weight=float(rng.choice([0.0, 1e-3, 1.0])),
                sim_batch_weight=float(rng.choice([0.0, 1e-3, 1.0])), u_n_prototypes=int(rng.choice([2, 7])),
                i_n_prototypes=int(rng.choice([3, 9])), u_sim_proto_weight=1e-3, u_sim_batch_weight=1e-3,
                i_sim_proto_weight=1e-3, i_sim_batch_weight=1e-3, n_anchors=int(rng.choice([2, 10])),
                delta_exc=float(rng.choice([0.0, 1e-2])), delta_inc=float(rng.choice([0.0, 1e-2])))
    if alg != 'mf':
        conf['embedding_dim'] = int(rng.choice([2, 7, 16, 30, 64]))
        conf['train_batch_size'] = int(rng.choice([7, 64, 128, 1000]))   # (autograd models: keep the run in seconds)
    desc = {k: conf[k] for k in ('embedding_dim', 'train_batch_size', 'neg_train', 'rec_loss', 'optimizer', 'train_neg_strategy',
                                 'eval_batch_size', 'use_user_bias', 'use_item_bias', 'use_global_bias', 'n_epochs')}
    desc.update(alg=alg, n_users=n_users, n_items=n_items, nnz=nnz, groups=groups)
    try:
        best, test, conf2 = run_train_val_test(AlgorithmsEnum[alg], DatasetsEnum.ml100k, dict(conf))
        bad = []
        for name, d in (('val', best), ('test', test)):
            for k, v in d.items():
                if k in ('best_epoch',):
                    continue
                if not np.isfinite(v) or (('@' in k) and not (0.0 <= v <= 1.0 + 1e-9)):
                    bad.append((name, k, v))
        if not os.path.isfile(os.path.join(conf2['model_path'], 'model.pth')):
            bad.append(('model.pth missing',))
        return (not bad), dict(desc, bad=bad[:4])
    except Exception as e:   # noqa: BLE001
        import traceback
        tb = traceback.format_exc().strip().splitlines()
        return False, dict(desc, raised=f'{type(e).__name__}: {e}'[:300], where=tb[-3:] if len(tb) >= 3 else tb)
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    root = tempfile.mkdtemp(prefix='hsk_stress_')
    t_end = time.time() + budget
    n = bad = 0
    devnull = open(os.devnull, 'w')
    try:
        while time.time() < t_end:
            out = sys.stdout
            sys.stdout = devnull            # the experiment helpers print their progress
            try:
                ok, desc = one_case(rng, root)
            finally:
                sys.stdout = out
            n += 1
            if not ok:
                bad += 1
                print('FAIL', desc, flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    print(f'{n} runs, {bad} failures', flush=True)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
