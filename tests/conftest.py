import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("SGAN_POISON_EMPTY"):
        # debugging aid: every torch.empty / empty_like of a floating dtype comes back filled with NaN, so a kernel that reads memory
        # nobody wrote shows up as NaNs in its result instead of as whatever the allocator's last tenant left there
        import torch
        real_empty, real_like = torch.empty, torch.empty_like

        def empty(*a, **k):
            t = real_empty(*a, **k)
            return t.fill_(float("nan")) if t.is_floating_point() else t

        def empty_like(*a, **k):
            t = real_like(*a, **k)
            return t.fill_(float("nan")) if t.is_floating_point() else t
        torch.empty, torch.empty_like = empty, empty_like


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def built_lib():
    """Path of libsgan_hip.so; built first (hipcc cross-compiles gfx950 without a GPU) when a fresh checkout lacks it."""
    from supervised_gan_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    assert os.path.exists(_lib.LIB_PATH), _lib.LIB_PATH
    return _lib.LIB_PATH
