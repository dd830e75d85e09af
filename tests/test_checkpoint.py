"""model.pth / conf.yml round trip with the reference: the on-disk formats on the output side of
the path (algorithms/base_classes.py:156-165, conf/conf_parser.py:46-51 of the reference).

tests/golden/g7_checkpoint/ is a run directory exactly as the reference leaves it (written by its own
save_model_to_path / save_yaml in oracle/gen_golden.py::gen_g7) plus the logits of the reloaded reference model."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN as GOLDEN_DIR

CKPT = os.path.join(GOLDEN_DIR, 'g7_checkpoint')
REF = '/root/reference'


def _expected():
    with np.load(os.path.join(CKPT, 'expected.npz')) as z:
        return {k: z[k] for k in z.files}


def _conf():
    from hassaku_amd.conf.conf_parser import parse_conf_file
    return parse_conf_file(os.path.join(CKPT, 'conf.yml'))


def test_reference_conf_yml_parses_and_is_a_fixed_point_of_parse_conf():
    from hassaku_amd.algorithms.algorithms_utils import AlgorithmsEnum
    from hassaku_amd.conf.conf_parser import parse_conf
    from hassaku_amd.data.data_utils import DatasetsEnum
    conf = _conf()
    assert conf['alg'] == 'mf' and conf['dataset'] == 'ml1m' and conf['embedding_dim'] == 24
    assert conf['running_settings']['seed'] == 64 and conf['wd'] == 4e-5
    before = {k: (dict(v) if isinstance(v, dict) else v) for k, v in conf.items()}
    conf['device'] = 'cuda'                    # the only key a user must change: there is no CPU trainer here
    after = parse_conf(conf, AlgorithmsEnum[conf['alg']], DatasetsEnum[conf['dataset']])
    for k, v in before.items():                # every default the reference filled in is accepted unchanged
        if k not in ('device', 'time_run', 'model_path'):
            assert after[k] == v, k


def test_reference_model_pth_has_the_keys_and_shapes_of_our_state_dict():
    from hassaku_amd.algorithms.sgd_alg import SGDMatrixFactorization
    conf = _conf()
    sd = torch.load(os.path.join(CKPT, 'model.pth'), map_location='cpu')
    ours = SGDMatrixFactorization.build_from_conf(conf, types.SimpleNamespace(n_users=37, n_items=53)).state_dict()
    assert list(sd.keys()) == list(ours.keys())
    for k in sd:
        assert sd[k].shape == ours[k].shape and sd[k].dtype == ours[k].dtype, k
    ex = _expected()
    for k in sd:
        assert np.array_equal(sd[k].numpy(), ex['sd.' + k]), k


@pytest.mark.gpu
def test_reference_checkpoint_loads_and_scores_like_the_reference():
    from hassaku_amd.algorithms.sgd_alg import SGDMatrixFactorization
    conf, ex = _conf(), _expected()
    model = SGDMatrixFactorization.build_from_conf(conf, types.SimpleNamespace(n_users=37, n_items=53)).to('cuda')
    model.load_model_from_path(CKPT)
    for k, v in model.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), ex['sd.' + k]), k
    out = model.predict(torch.from_numpy(ex['u_idx']).cuda(), torch.from_numpy(ex['i_idx']).cuda()).cpu().numpy()
    assert np.abs(out - ex['logits']).max() <= 1e-5 * np.abs(ex['logits']).max()


@pytest.mark.skipif(not os.path.isdir(REF), reason='needs the reference checkout (authoring container only)')
def test_our_checkpoint_loads_in_the_reference(tmp_path):
    """The other direction: a run directory written by this package is read back by the reference's own classes."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import gen_golden
    gen_golden.import_reference()               # reference modules + the documented stand-ins
    try:
        from algorithms.sgd_alg import SGDMatrixFactorization as RefMF
        from conf.conf_parser import parse_conf_file as ref_parse_conf_file
        from hassaku_amd.algorithms.sgd_alg import SGDMatrixFactorization
        from hassaku_amd.conf.conf_parser import save_yaml
        conf = _conf()
        ds = types.SimpleNamespace(n_users=37, n_items=53)
        torch.manual_seed(5)
        ours = SGDMatrixFactorization.build_from_conf(conf, ds)
        with torch.no_grad():
            for p in ours.parameters():
                p.add_(torch.randn_like(p) * 0.2)
        ours.save_model_to_path(str(tmp_path))
        save_yaml(str(tmp_path), conf)
        ref_conf = ref_parse_conf_file(os.path.join(str(tmp_path), 'conf.yml'))
        assert ref_conf == conf
        ref = RefMF.build_from_conf(ref_conf, ds)
        ref.load_model_from_path(str(tmp_path))
        for (k0, v0), (k1, v1) in zip(ours.state_dict().items(), ref.state_dict().items()):
            assert k0 == k1 and torch.equal(v0, v1), k0
    finally:
        for p in (REF,):
            while p in sys.path:
                sys.path.remove(p)
        for name in [n for n in sys.modules if n.split('.')[0] in
                     ('algorithms', 'conf', 'data', 'eval', 'train', 'utilities', 'explanations', 'hyper_search',
                      'wandb', 'ray', 'gdown')]:
            del sys.modules[name]
