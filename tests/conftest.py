import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session', autouse=True)
def _remove_checkpoint_fixtures(tmp_path_factory):
    """the checkpoint files the tests write (tests/fake_yolov5.save_checkpoint: up to 280 MB each at the x6 widths) go when the
    session ends -- pytest keeps the temporary directories of the last three sessions, and these would stay with them"""
    yield
    try:
        for p in tmp_path_factory.getbasetemp().rglob('*.pt'):
            try:
                p.unlink()
            except OSError:
                pass
    except Exception:
        pass
