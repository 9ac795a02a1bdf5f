import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests skip (instead of failing) on a box without a device -- unless they were asked
    for explicitly with ``-m gpu``: the GPU tier must fail loudly when the device is missing."""
    if 'gpu' in (config.getoption('-m') or '') and 'not gpu' not in config.getoption('-m'):
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='needs a GPU (torch.cuda.is_available() is False)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
