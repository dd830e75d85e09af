"""`get_dataloader` factory and the dataset registry (data/data_utils.py:36-47,317-375).

The reference's download / k-core / split preprocessing needs the network and is out of scope; processed
datasets in the 5-CSV layout (or hassaku_amd.data.synthetic) are consumed as they are.  `ml100k` is
added to the registry: BASELINE.json names it and the reference ships a processor but no enum member.
"""
import enum
import logging

from torch.utils.data import DataLoader

from hassaku_amd.data.dataloader import NegativeSampler, TrainDataLoader
from hassaku_amd.data.dataset import FullEvalDataset, TrainRecDataset


class DatasetsEnum(enum.Enum):
    ml1m = enum.auto()
    ml10m = enum.auto()
    amazonvid2018 = enum.auto()
    lfm2b2020 = enum.auto()
    deliveryherosg = enum.auto()
    lfm2bdemobias = enum.auto()
    deezer = enum.auto()
    ml100k = enum.auto()


class _EvalLoader:
    """Evaluation 'loader' for the HIP evaluator: carries the dataset and the batch size; iterating it yields
    the reference's dense (u, arange(I), labels) batches for generic algorithms."""

    def __init__(self, dataset: FullEvalDataset, batch_size: int, num_workers: int = 0):
        self.dataset, self.batch_size, self.num_workers = dataset, batch_size, num_workers

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        return iter(DataLoader(self.dataset, batch_size=self.batch_size, num_workers=0))


def get_dataloader(conf: dict, split_set: str):
    running = conf['running_settings']
    if split_set == 'train':
        dataset = TrainRecDataset(data_path=conf['dataset_path'])
        sampler = NegativeSampler(train_dataset=dataset, n_neg=conf['neg_train'],
                                  neg_sampling_strategy=conf['train_neg_strategy'])
        loader = TrainDataLoader(sampler, dataset, batch_size=conf['train_batch_size'], shuffle=True,
                                 num_workers=running.get('train_n_workers', 0), device=conf.get('device', 'cuda'))
        logging.info('Built Train DataLoader batch_size=%d', conf['train_batch_size'])
        return loader
    if split_set in ('val', 'test'):
        loader = _EvalLoader(FullEvalDataset(data_path=conf['dataset_path'], split_set=split_set),
                             batch_size=conf['eval_batch_size'], num_workers=running.get('eval_n_workers', 0))
        logging.info('Built %s DataLoader batch_size=%d', split_set, conf['eval_batch_size'])
        return loader
    raise ValueError(f"split_set value '{split_set}' is invalid! Please choose from [train, val, test]")
