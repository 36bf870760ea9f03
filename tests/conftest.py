import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


def pytest_runtest_setup(item):
    """Several GPU tests start child processes on the same card (ranks of a data-parallel run, bench.py and its
    native-fp32 leg).  This process's caching allocator never returns memory by itself, and after a few full-size tests it
    can hold most of the 288 GB -- a child then dies with 'HIP out of memory ... 0 bytes free' (seen in round 6).  Before
    every GPU test: if more than 64 GiB are cached, give the cache back."""
    if item.get_closest_marker('gpu') is None or 'torch' not in sys.modules:
        return
    import torch
    if torch.cuda.is_available() and torch.cuda.is_initialized() and torch.cuda.memory_reserved() > (64 << 30):
        import gc
        gc.collect()
        torch.cuda.empty_cache()


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


# Achieved parity figures (worst errors inside the tolerances) are collected here by the tests and printed in the
# terminal summary, so they show up in the tail of a `pytest -q` log even when everything passes.
ACHIEVED = []


FLIPS = {}        # fixture name -> "k of n ReLU decisions differ" (filled by the ReLU-decision test, read at summary time)


def record_achieved(line):
    """line: a string, or a callable evaluated when the summary is printed (so a figure measured by a later test can
    stand beside it)"""
    ACHIEVED.append(line if callable(line) else str(line))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if ACHIEVED:
        terminalreporter.write_sep('-', 'achieved parity figures (inside the asserted tolerances)')
        for line in ACHIEVED:
            terminalreporter.write_line(line() if callable(line) else line)
