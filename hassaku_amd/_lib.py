"""ctypes binding of libhassaku_hip.so (the C ABI declared in include/hassaku_hip.h).

The library is the product: there is no CPU or PyTorch fallback behind these calls.  If the shared
object is missing, or a compute entry point is called without a GPU, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('HSK_LIB_PATH') or os.path.join(_HERE, 'libhassaku_hip.so')   # override: A/B builds

_lib = None


class HskBprmfState(ctypes.Structure):
    """Mirror of `struct hsk_bprmf_state` (include/hassaku_hip.h)."""
    _fields_ = [
        ('user_emb', c_void_p), ('item_emb', c_void_p), ('item_bias', c_void_p),
        ('user_bias', c_void_p), ('global_bias', c_void_p),
        ('m_user_emb', c_void_p), ('v_user_emb', c_void_p),
        ('m_item_emb', c_void_p), ('v_item_emb', c_void_p),
        ('m_item_bias', c_void_p), ('v_item_bias', c_void_p),
        ('m_user_bias', c_void_p), ('v_user_bias', c_void_p),
        ('m_global_bias', c_void_p), ('v_global_bias', c_void_p),
        ('n_users', c_int64), ('n_items', c_int64), ('dim', c_int64),
        ('lr', c_double), ('beta1', c_double), ('beta2', c_double), ('eps', c_double), ('wd', c_double),
        ('step', c_int64),
        ('csr_indptr', c_void_p), ('csr_indices', c_void_p),
        ('coo_user', c_void_p), ('coo_item', c_void_p),
        ('nnz', c_int64),
        ('seed', c_uint64),
        ('workspace', c_void_p), ('workspace_bytes', c_int64),
        ('max_batch', c_int64), ('max_cols', c_int64),
        ('lazy_users', c_int32), ('timing_mask', c_int32),
        ('timing', c_void_p), ('aux', c_void_p),
        ('timing_every', c_int32), ('loss_kind', c_int32),
        ('opt_kind', c_int32), ('lazy_items', c_int32),
        ('ssm_log_adjust', c_double),
        ('alias_prob', c_void_p), ('alias_idx', c_void_p),
        ('graph_chunk', c_int32), ('catchup_apart', c_int32),
        ('loss_out', c_void_p), ('status', c_void_p),
        ('flush_every', c_int32), ('ws_sharded', c_int32),
        # library scratch (zero-initialised by ctypes, never written from Python)
        ('frozen_hyper', c_double * 5), ('frozen_opt', c_int32), ('frozen_valid', c_int32),
        ('timing_now', c_int32), ('reserved3', c_int32),
    ]


class HskBprmfShard(ctypes.Structure):
    """Mirror of `struct hsk_bprmf_shard` (include/hassaku_hip.h)."""
    _fields_ = [
        ('base', HskBprmfState),
        ('world', c_int32), ('rank', c_int32),
        ('n_users_global', c_int64), ('n_items_global', c_int64),
        ('item_lo', c_int64), ('capacity', c_int64), ('entry_cap', c_int64),
        ('shard_ws', c_void_p), ('shard_ws_bytes', c_int64),
        ('rows_send', c_void_p), ('rows_all', c_void_p),
        ('dU_all', c_void_p), ('grads_mine', c_void_p),
        ('s0', c_void_p), ('gsum', c_void_p),
        ('ssm_send', c_void_p), ('ssm_all', c_void_p),
        ('cur_batch', c_int64), ('cur_cols', c_int64),
        ('cur_set', c_int32), ('phase', c_int32),
    ]


# name -> (restype, argtypes); every symbol of include/hassaku_hip.h
SIGNATURES = {
    'hsk_version': (c_int, []),
    'hsk_last_error': (c_char_p, []),
    'hsk_device_info': (c_int, [POINTER(c_int32), POINTER(c_int32), c_char_p, c_int32]),
    'hsk_mf_scores': (c_int, [c_void_p] * 5 + [c_int64] * 3 + [c_void_p, c_void_p, c_int64, c_int64,
                                                              c_void_p, c_void_p, c_void_p]),
    'hsk_bpr_loss_grad': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'hsk_rec_loss_grad': (c_int, [c_int32, c_void_p, c_int64, c_int64, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    'hsk_mf_backward': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                                c_int64, c_void_p] + [c_void_p] * 5 + [c_void_p, c_void_p]),
    'hsk_embedding_gather': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    'hsk_embedding_backward_ws_bytes': (c_int64, [c_int64, c_int64]),
    'hsk_embedding_backward': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64,
                                       c_void_p, c_void_p]),
    'hsk_adamw_dense': (c_int, [c_void_p] * 4 + [c_int64] + [c_double] * 5 + [c_int64, c_void_p]),
    'hsk_opt_dense': (c_int, [c_int] + [c_void_p] * 4 + [c_int64] + [c_double] * 5 + [c_int64, c_void_p]),
    'hsk_sample_negatives_uniform': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                             c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    'hsk_sample_negatives_alias': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                           c_int64, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p]),
    'hsk_bprmf_workspace_bytes': (c_int64, [c_int64] * 5),
    'hsk_bprmf_flush_cadence': (c_int32, [POINTER(HskBprmfState), c_int32, c_int64]),
    'hsk_shard_base_workspace_bytes': (c_int64, [c_int64] * 5),
    'hsk_synth_degrees': (c_int, [c_int64, c_int32, c_int32, c_uint64, c_void_p, c_void_p]),
    'hsk_synth_fill': (c_int, [c_int64, c_int64, c_int32, c_int32, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'hsk_bprmf_init_workspace': (c_int, [POINTER(HskBprmfState), c_void_p]),
    'hsk_bprmf_train_step': (c_int, [POINTER(HskBprmfState), c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'hsk_bprmf_train_step_sampled': (c_int, [POINTER(HskBprmfState), c_void_p, c_int64, c_int64, c_int64,
                                             c_void_p]),
    'hsk_bprmf_train_steps': (c_int, [POINTER(HskBprmfState), c_void_p, c_int64, c_int64, c_int64, c_int64,
                                      c_void_p]),
    'hsk_bprmf_last_sort': (c_int, [POINTER(HskBprmfState), c_int64, c_void_p, c_void_p, c_void_p]),
    'hsk_bprmf_graph_replays': (c_int64, [POINTER(HskBprmfState)]),
    'hsk_bprmf_batch_columns': (c_int64, [POINTER(HskBprmfState), c_int64, c_int64]),
    'hsk_eval_set_arith': (None, [c_int]),
    'hsk_timing_create': (c_void_p, []),
    'hsk_timing_destroy': (None, [c_void_p]),
    'hsk_timing_collect': (c_int, [c_void_p, POINTER(c_double), POINTER(c_int64)]),
    'hsk_aux_create': (c_void_p, []),
    'hsk_aux_destroy': (None, [c_void_p]),
    'hsk_bprmf_hint_next': (c_int, [POINTER(HskBprmfState), c_void_p, c_int64, c_int64, c_int64]),
    'hsk_bprmf_hint_after_run': (c_int, [POINTER(HskBprmfState), c_void_p, c_int64, c_int64, c_int64]),
    'hsk_bprmf_hint_after_run_n': (c_int, [POINTER(HskBprmfState), c_void_p, c_int64, c_int64, c_int64, c_int64]),
    'hsk_bprmf_pipelined_steps': (c_int64, [POINTER(HskBprmfState)]),
    'hsk_bprmf_set_pipeline': (None, [c_int]),
    'hsk_shard_workspace_bytes': (c_int64, [c_int64] * 4),
    'hsk_shard_init': (c_int, [POINTER(HskBprmfShard), c_void_p]),
    'hsk_shard_prepare': (c_int, [POINTER(HskBprmfShard), c_void_p, c_int64, c_int64, c_int64, c_int32, c_void_p]),
    'hsk_shard_discard': (c_int, [POINTER(HskBprmfShard), c_int32, c_void_p]),
    'hsk_shard_pack': (c_int, [POINTER(HskBprmfShard), c_int64, c_int64, c_int32, c_void_p]),
    'hsk_shard_pos_scores': (c_int, [POINTER(HskBprmfShard), c_void_p]),
    'hsk_shard_forward': (c_int, [POINTER(HskBprmfShard), c_void_p]),
    'hsk_shard_pos_fix': (c_int, [POINTER(HskBprmfShard), c_void_p]),
    'hsk_shard_apply_items': (c_int, [POINTER(HskBprmfShard), c_void_p]),
    'hsk_shard_apply_users': (c_int, [POINTER(HskBprmfShard), c_void_p]),
    'hsk_shard_flush': (c_int, [POINTER(HskBprmfShard), c_void_p]),
    'hsk_shard_last_batch': (c_int, [POINTER(HskBprmfShard), c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'hsk_rccl_available': (c_int, []),
    'hsk_rccl_unique_id': (c_int, [c_void_p]),
    'hsk_hostcoll_unique_id': (c_int, [c_int64, c_void_p]),
    'hsk_shard_rt_create': (c_void_p, [c_int32, c_int32, c_void_p]),
    'hsk_shard_rt_create_with': (c_void_p, [c_int32, c_int32, c_void_p]),
    'hsk_shard_rt_backend': (c_char_p, [c_void_p]),
    'hsk_shard_rt_destroy': (None, [c_void_p]),
    'hsk_shard_step': (c_int, [POINTER(HskBprmfShard), c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                               c_void_p]),
    'hsk_shard_rt_flush': (c_int, [POINTER(HskBprmfShard), c_void_p, c_void_p]),
    'hsk_shard_rt_cur_set': (c_int, [c_void_p]),
    'hsk_shard_rt_discard_prefetch': (c_int, [POINTER(HskBprmfShard), c_void_p, c_void_p]),
    'hsk_bprmf_flush': (c_int, [POINTER(HskBprmfState), c_void_p]),
    'hsk_bprmf_last_batch': (c_int, [POINTER(HskBprmfState), c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    'hsk_mf_eval_topk': (c_int, [c_void_p] * 5 + [c_int64] * 3 + [c_void_p, c_int64, c_int64, c_int64,
                                                                 c_void_p, c_void_p, c_int64,
                                                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'hsk_mf_eval_planes_bytes': (c_int64, [c_int64] * 3),
    'hsk_mf_eval_topk_planes': (c_int, [c_void_p] * 5 + [c_int64] * 3 + [c_void_p, c_int64, c_int64, c_int64,
                                                                        c_void_p, c_void_p, c_int64,
                                                                        c_void_p, c_void_p, c_int64,
                                                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    'hsk_mf_eval_fused_ws_bytes': (c_int64, [c_int64] * 3),
    'hsk_mf_eval_fused_ws_bytes_dim': (c_int64, [c_int64] * 4),
    'hsk_mf_eval_topk_fused': (c_int, [c_void_p] * 5 + [c_int64] * 3 + [c_void_p, c_int64, c_int64, c_int64,
                                                                       c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    'hsk_topk_dense': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    'hsk_topk_merge': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    'hsk_rank_metrics': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                 POINTER(c_int32), c_int32, c_void_p, c_void_p]),
}


def open_library(path):
    """ctypes handle of one build of the library with every prototype attached."""
    if not os.path.isfile(path):
        raise RuntimeError(
            f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C hassaku_amd/csrc`.  There is no CPU fallback for the HIP path.')
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def load():
    """Load the shared library once and attach the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        _lib = open_library(LIB_PATH)
    return _lib


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = load().hsk_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'hassaku_hip {what} failed (code {rc}): {msg}')


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError('hassaku_amd: no HIP device visible; the MI355X path has no CPU fallback')
