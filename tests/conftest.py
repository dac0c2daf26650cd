import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(autouse=True)
def _restore_pooling_cfg():
    """tests that switch the product's top-level pooling options put them back"""
    yield
    mod = sys.modules.get('lang2seg_amd.model.config')
    if mod is not None:
        mod.cfg.POOLING_MODE = 'crop'; mod.cfg.POOLING_ALIGN = False; mod.cfg.RESNET.MAX_POOL = False; mod.cfg.RESNET.FIXED_BLOCKS = 1
