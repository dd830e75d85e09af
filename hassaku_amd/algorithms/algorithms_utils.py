"""Name -> class registry (algorithms/algorithms_utils.py:12-30).  The slot on the hot path (mf) and its bias-only
sibling (sgdbias) are filled; the reference's other sixteen algorithms are out of scope
(SURVEY.md section 2)."""
from enum import Enum

from hassaku_amd.algorithms.sgd_alg import SGDBaseline, SGDMatrixFactorization


class AlgorithmsEnum(Enum):
    mf = SGDMatrixFactorization
    sgdbias = SGDBaseline
