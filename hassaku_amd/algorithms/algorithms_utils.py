"""Name -> class registry (algorithms/algorithms_utils.py:12-30).  Filled: the slot on the hot path (mf), its
bias-only sibling (sgdbias) and the anchor / prototype models that share the embedding gather; the
reference's other twelve algorithms are out of scope."""
from enum import Enum

from hassaku_amd.algorithms.proto_alg import ACF, IProtoMF, UIProtoMF, UProtoMF
from hassaku_amd.algorithms.sgd_alg import SGDBaseline, SGDMatrixFactorization


class AlgorithmsEnum(Enum):
    mf = SGDMatrixFactorization
    sgdbias = SGDBaseline
    uprotomf = UProtoMF
    iprotomf = IProtoMF
    uiprotomf = UIProtoMF
    acf = ACF
