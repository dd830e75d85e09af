#!/usr/bin/env python3
"""Command line of the MI355X path.  Same flags, defaults and run types as the reference's driver script
(run_experiment.py:8-36 there), so existing invocations keep working:

    python run_experiment.py -a mf -d ml1m -c conf.yml [-t train_val|test|train_val_test] [--log LEVEL]

Launched under `python -m torch.distributed.run --nproc-per-node N ...` it trains with one process per GPU
(hassaku_amd/dist.py); nothing else changes on the command line.
"""
import argparse
import logging
import sys

from hassaku_amd import experiment_helper as helper
from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
from hassaku_amd.data.data_utils import DatasetsEnum

RUN_TYPES = {                      # -t value -> what it runs
    'train_val': helper.run_train_val,
    'test': helper.run_test,
    'train_val_test': helper.run_train_val_test,
}


def _flags():
    """(names, argparse keywords) for every flag of the reference's CLI."""
    yield ('-a', '--algorithm'), dict(choices=sorted(m.name for m in AlgorithmsEnum), metavar='ALG',
                                      help='registry name of the model (%(choices)s)')
    yield ('-d', '--dataset'), dict(choices=sorted(m.name for m in DatasetsEnum), default='ml1m', metavar='DATASET',
                                    help='registry name of the dataset, default %(default)s')
    yield ('-c', '--conf_path'), dict(metavar='FILE', help='YAML (or JSON) file with the experiment configuration')
    yield ('-t', '--run_type'), dict(choices=sorted(RUN_TYPES), default='train_val_test',
                                     help='which phases to run, default %(default)s')
    yield ('--log',), dict(default='WARNING', metavar='LEVEL', help='logging level, default %(default)s')


def main(argv=None) -> int:
    cli = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for names, kw in _flags():
        cli.add_argument(*names, type=str, **kw)
    opts = cli.parse_args(argv)
    logging.basicConfig(level=opts.log)
    RUN_TYPES[opts.run_type](AlgorithmsEnum[opts.algorithm], DatasetsEnum[opts.dataset], opts.conf_path)
    return 0


if __name__ == '__main__':
    sys.exit(main())
