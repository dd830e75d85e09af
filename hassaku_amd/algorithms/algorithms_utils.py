"""Name -> class registry (algorithms/algorithms_utils.py:12-30).  Only the slot on the hot path is filled;
the reference's other seventeen algorithms are out of scope."""
from enum import Enum

from hassaku_amd.algorithms.sgd_alg import SGDMatrixFactorization


class AlgorithmsEnum(Enum):
    mf = SGDMatrixFactorization
