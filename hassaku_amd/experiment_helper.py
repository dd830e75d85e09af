"""Train / validate / test orchestration -- the reference's experiment_helper.py:18-130 for the SGD family.

Same three entry points and the same conf contract; wandb is optional (used only if importable and
`running_settings.use_wandb` is true).  Unlike the reference's run_test (which evaluates on the CPU because
it passes no device, experiment_helper.py:116-117), the test split is scored on the HIP device.
"""
import typing

from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
from hassaku_amd.algorithms.base_classes import SGDBasedRecommenderAlgorithm
from hassaku_amd.conf.conf_parser import parse_conf, parse_conf_file, save_yaml
from hassaku_amd.data.data_utils import DatasetsEnum, get_dataloader
from hassaku_amd.eval.eval import FullEvaluator, evaluate_recommender_algorithm
from hassaku_amd.train.rec_losses import RecommenderSystemLossesEnum
from hassaku_amd.train.trainer import Trainer
from hassaku_amd.utilities.utils import reproducible


def maybe_init_distributed(conf) -> int:
    """One process per GPU (torchrun / torch.distributed.run): join the RCCL process group and pin this rank to
    its GPU.  Returns the world size (1 when launched as a plain process)."""
    import os
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 1
    if not dist.is_initialized():
        local = int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = conf.get('running_settings', {}).get('dist_backend', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    return dist.get_world_size()


def _wandb(conf):
    if not conf['running_settings'].get('use_wandb'):
        return None
    try:
        import wandb
        return wandb
    except ImportError:
        print('use_wandb is set but wandb is not installed: logging to stdout only')
        conf['running_settings']['use_wandb'] = False
        return None


def run_train_val(alg: AlgorithmsEnum, dataset: DatasetsEnum, conf: typing.Union[str, dict]):
    print(f'Starting Train-Val\nAlgorithm is {alg.name} - Dataset is {dataset.name}')
    if isinstance(conf, str):
        conf = parse_conf_file(conf)
    conf = parse_conf(conf, alg, dataset)
    world = maybe_init_distributed(conf)
    if world > 1:
        import torch.distributed as dist
        if dist.get_rank() != 0:
            conf['running_settings']['use_wandb'] = False
        # every rank must write into the SAME run folder: rank 0's time id wins
        box = [conf['model_path'], conf['time_run']]
        dist.broadcast_object_list(box, src=0)
        conf['model_path'], conf['time_run'] = box
    wandb = _wandb(conf)
    if wandb is not None:
        wandb.init(config=conf, tags=[alg.name, dataset.name], name=conf['time_run'], job_type='train/val')
    reproducible(conf['running_settings']['seed'])
    if not issubclass(alg.value, SGDBasedRecommenderAlgorithm):
        raise ValueError(f'Training for {alg.value} has been not implemented')
    train_loader = get_dataloader(conf, 'train')
    val_loader = get_dataloader(conf, 'val')
    model = alg.value.build_from_conf(conf, train_loader.dataset)
    rec_loss = RecommenderSystemLossesEnum[conf['rec_loss']].value.build_from_conf(conf, train_loader.dataset)
    trainer = Trainer(model, train_loader, val_loader, rec_loss, conf)
    metrics_values = trainer.fit()
    if world == 1 or trainer.comm.rank == 0:
        save_yaml(conf['model_path'], conf)
    if wandb is not None:
        wandb.finish()
    return metrics_values, conf


def run_test(alg: AlgorithmsEnum, dataset: DatasetsEnum, conf: typing.Union[str, dict]):
    print(f'Starting Test\nAlgorithm is {alg.name} - Dataset is {dataset.name}')
    if isinstance(conf, str):
        conf = parse_conf_file(conf)
    wandb = _wandb(conf)
    if wandb is not None:
        wandb.init(config=conf, tags=[alg.name, dataset.name], name=conf['time_run'], job_type='test', reinit=True)
    test_loader = get_dataloader(conf, 'test')
    model = alg.value.build_from_conf(conf, test_loader.dataset).to(conf.get('device', 'cuda'))
    model.load_model_from_path(conf['model_path'])
    evaluator = FullEvaluator(aggr_by_group=True, n_groups=test_loader.dataset.n_user_groups,
                              user_to_user_group=test_loader.dataset.user_to_user_group)
    metrics_values = evaluate_recommender_algorithm(model, test_loader, evaluator, conf.get('device', 'cuda'),
                                                    verbose=conf['running_settings']['batch_verbose'])
    if wandb is not None:
        wandb.log(metrics_values, step=0)
        wandb.finish()
    return metrics_values


def run_train_val_test(alg: AlgorithmsEnum, dataset: DatasetsEnum, conf_path: typing.Union[str, dict]):
    print(f'Starting Train-Val-Test\nAlgorithm is {alg.name} - Dataset is {dataset.name}')
    metrics_values, conf = run_train_val(alg, dataset, conf_path)
    test_metrics = run_test(alg, dataset, conf)
    return metrics_values, test_metrics, conf
