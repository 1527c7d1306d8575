import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# A fatal signal in the test process (round 5: one full-suite run ended in one, its log lost to a `tail`) must leave a trace: every
# thread's Python stack goes to gpurun_out/faulthandler_<pid>.txt (merged back from the GPU box) and to stderr; processes the
# tests spawn inherit PYTHONFAULTHANDLER=1 and dump to their own stderr.
_FAULT_LOG = None


def _enable_faulthandler():
    global _FAULT_LOG
    import faulthandler
    os.environ.setdefault("PYTHONFAULTHANDLER", "1")
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        _FAULT_LOG = open(os.path.join(out, "faulthandler_%d.txt" % os.getpid()), "w")
        faulthandler.enable(file=_FAULT_LOG, all_threads=True)
    except OSError:
        faulthandler.enable(all_threads=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _enable_faulthandler()


def pytest_unconfigure(config):
    # a clean exit leaves no (empty) dump file behind
    global _FAULT_LOG
    if _FAULT_LOG is not None:
        import faulthandler
        faulthandler.disable()
        name = _FAULT_LOG.name
        _FAULT_LOG.close()
        _FAULT_LOG = None
        try:
            if os.path.getsize(name) == 0:
                os.remove(name)
        except OSError:
            pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
