"""The driver's contract with bench.py: one JSON line, the agreed keys, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--workload', 'ml100k', '--steps', '6',
                          '--warmup', '2', '--cpu-budget', '1', '--only'], cwd=REPO, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['metric'] == 'BPR triplets/sec' and d['unit'] == 'triplets/s' and d['higher_is_better'] is True
    assert d['n_gpus'] == 1 and d['steps'] == 6 and d['warmup'] == 2 and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert d['value'] > 0 and abs(d['value'] * d['ms_per_step'] * 1e-3 - 128 * 1) < 1e-6 * 128      # B=128, N=1
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    # the ml100k tables (0.4 MB) are cache-resident: the bound is the cache fabric and the peak the guide's measured
    # ceiling for random row gathers from the Infinity Cache; the fraction of the 8 TB/s HBM spec rides along
    assert r['bound'] == 'infinity-cache' and r['unit'] == 'GB/s' and r['peak'] == 8600.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert abs(r['frac_of_hbm_spec_8000'] - r['achieved'] / 8000.0) < 1e-12
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['cpu_model']
    # configs[0], 20 warm-up + 200 timed steps, with 0 and with 4 loader workers
    assert [leg['train_n_workers'] for leg in c['legs']] == [0, 4]
    assert all(leg['steps'] >= 200 and leg['warmup'] == 20 and leg['value'] > 0 for leg in c['legs'])


@pytest.mark.gpu
def test_cfg5_share_leg_runs_at_a_scaled_shape(monkeypatch):
    """bench.run_cfg5 (the `workloads.cfg5_shard` leg: rank 3 of 8 of BASELINE configs[4] on its own, collectives stubbed by
    dist.LoopbackComm) on tables scaled down 500 x: the leg's code path, keys and bookkeeping -- so that a regression shows
    here and not as an `error` entry in the driver's bench line."""
    import torch
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.setitem(bench.CFG5, 'U', 200_000)
    monkeypatch.setitem(bench.CFG5, 'I', 80_000)
    monkeypatch.setitem(bench.CFG5, 'B', 64)
    x = bench.run_cfg5(torch.device('cuda', 0), 3, 2)
    assert x['ranks_running'] == 1 and x['n_gpus_of_the_job'] == 8 and x['ms_per_step'] > 0
    assert x['roofline']['bound'] == 'hbm' and x['roofline']['launches'] == 3 and x['roofline']['achieved'] > 0
    assert set(x['stage_us_per_step']) == {'fwd', 'item', 'user'}
    assert x['roofline']['kept_entries_per_step'] > 64 * 8 * 200 // 16       # about an eighth of the global batch's entries
    assert x['lazy_sweep']['sweep_ms'] > 0 and x['value_rank_share'] > 0 and np.isfinite(x['loss_last_step_local_share'])
