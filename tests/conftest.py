import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


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
