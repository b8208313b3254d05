import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


CALIB_KEYS = ('sensor2ego', 'ego2global', 'intrin', 'post_rot', 'post_tran', 'bda')


def golden_calib(g):
    return [g[k] for k in CALIB_KEYS]


def small_dhds_cfg():
    from dhd_amd import synthetic as syn
    cfg = syn.dhd_s_config()
    cfg['input_size'] = (64, 176)
    cfg['out_channels'] = 8
    return cfg


@pytest.fixture(scope='session')
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda', 0)
