#!/usr/bin/env python3
"""CLI with the reference's flags (run_experiment.py:8-36):
    python run_experiment.py -a mf -d ml1m -c conf.yml [-t train_val|test|train_val_test] [--log LEVEL]
"""
import argparse
import logging

from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
from hassaku_amd.data.data_utils import DatasetsEnum
from hassaku_amd.experiment_helper import run_test, run_train_val, run_train_val_test


def main():
    parser = argparse.ArgumentParser(description='Start an experiment')
    parser.add_argument('--algorithm', '-a', type=str, choices=[a.name for a in AlgorithmsEnum],
                        help='Recommender Systems Algorithm')
    parser.add_argument('--dataset', '-d', type=str, choices=[d.name for d in DatasetsEnum], default='ml1m',
                        help='Recommender Systems Dataset')
    parser.add_argument('--conf_path', '-c', type=str, help='Path to the .yml containing the configuration')
    parser.add_argument('--run_type', '-t', type=str, choices=['train_val', 'test', 'train_val_test'],
                        default='train_val_test')
    parser.add_argument('--log', type=str, default='WARNING')
    args = parser.parse_args()
    logging.basicConfig(level=args.log)
    alg, dataset = AlgorithmsEnum[args.algorithm], DatasetsEnum[args.dataset]
    runner = {'train_val': run_train_val, 'test': run_test, 'train_val_test': run_train_val_test}[args.run_type]
    runner(alg, dataset, args.conf_path)


if __name__ == '__main__':
    main()
