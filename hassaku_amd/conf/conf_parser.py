"""Configuration loading and defaulting with the reference's keys (conf/conf_parser.py:12-186).

A conf is a plain dict read from YAML or JSON.  `parse_conf` fills the same defaults the reference fills,
validates the same things, and creates `model_path` = <model_save_path>/<alg>-<dataset>/single_runs/<time id>
(or sweeps/<sweep_id>/...).  Written as a table of (key, default) pairs rather than a chain of ifs.
"""
import json
import os

import yaml

from hassaku_amd.utilities.utils import generate_id

# reference defaults (conf/conf_parser.py:12-29)
TOP_LEVEL_DEFAULTS = [('optimizing_metric', 'ndcg@10'), ('eval_batch_size', 64)]
RUNNING_DEFAULTS = [('seed', 64), ('use_wandb', True), ('eval_n_workers', 2), ('batch_verbose', False)]
SGD_DEFAULTS = [('neg_train', 4), ('train_neg_strategy', 'uniform'), ('train_batch_size', 64), ('n_epochs', 50),
                ('lr', 1e-3), ('wd', 0), ('optimizer', 'adam'), ('rec_loss', 'bce'), ('device', 'cpu')]
SGD_RUNNING_DEFAULTS = [('train_n_workers', 2)]
DEF_MODEL_SAVE_PATH = './saved_models'
OPTIMIZERS = ('adam', 'adagrad', 'adamw')
DEVICES = ('cpu', 'cuda')


def parse_conf_file(conf_path: str) -> dict:
    assert os.path.isfile(conf_path), f'Configuration File {conf_path} not found!'
    with open(conf_path, 'r') as fh:
        text = fh.read()
    try:
        conf = yaml.safe_load(text)
    except yaml.YAMLError:
        conf = json.loads(text)
    return conf


def save_yaml(conf_path: str, conf: dict):
    with open(os.path.join(conf_path, 'conf.yml'), 'w') as fh:
        yaml.dump(conf, fh)


def _fill(target: dict, defaults, added: list):
    for key, value in defaults:
        if key not in target:
            target[key] = value
            added.append(f'{key}={value}')


def parse_conf(conf: dict, alg, dataset) -> dict:
    """alg / dataset are members of AlgorithmsEnum / DatasetsEnum (only .name and alg.value are used)."""
    from hassaku_amd.algorithms.base_classes import SGDBasedRecommenderAlgorithm
    from hassaku_amd.train.rec_losses import RecommenderSystemLossesEnum

    assert 'data_path' in conf, 'Data path is missing from the configuration file'
    conf['alg'] = alg.name
    conf['time_run'] = generate_id()
    conf['dataset'] = dataset.name
    conf.setdefault('dataset_path', os.path.join(conf['data_path'], conf['dataset'], 'processed_dataset'))
    in_tune = bool(conf.get('_in_tune'))
    added = []

    if not in_tune:
        _fill(conf, [('model_save_path', DEF_MODEL_SAVE_PATH)], added)
        middle = f"sweeps/{conf['sweep_id']}" if 'sweep_id' in conf else 'single_runs'
        conf['model_path'] = os.path.join(conf['model_save_path'], f'{alg.name}-{dataset.name}', middle, conf['time_run'])
        os.makedirs(conf['model_path'], exist_ok=True)

    _fill(conf, TOP_LEVEL_DEFAULTS, added)
    running = conf.setdefault('running_settings', {})
    _fill(running, RUNNING_DEFAULTS, added)
    if in_tune:
        _fill(running, [('ray_verbose', 1)], added)

    if issubclass(alg.value, SGDBasedRecommenderAlgorithm):
        given = set(conf)
        _fill(conf, SGD_DEFAULTS, added)
        if 'n_epochs' in given:
            assert conf['n_epochs'] > 0, f"Number of epochs ({conf['n_epochs']}) should be positive"
        if 'optimizer' in given:
            assert conf['optimizer'] in OPTIMIZERS, f"Optimizer ({conf['optimizer']}) not implemented"
        if 'rec_loss' in given:
            assert conf['rec_loss'] in [loss.name for loss in RecommenderSystemLossesEnum], \
                f"Rec loss ({conf['rec_loss']}) not implemented"
        if 'device' in given:
            assert conf['device'] in DEVICES, f"Device ({conf['device']}) not available"
        if 'max_patience' in given:
            assert 0 < conf['max_patience'] < conf['n_epochs'], \
                f"Max patience {conf['max_patience']} should be between 0 and {conf['n_epochs']}"
        else:
            conf['max_patience'] = conf['n_epochs'] - 1
            added.append(f"max_patience={conf['max_patience']}")
        _fill(running, SGD_RUNNING_DEFAULTS, added)

    print('Added these default parameters: ', ', '.join(added))
    return conf
