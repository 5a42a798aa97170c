import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no HIP device is visible and -m gpu was not forced."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Worst errors the parity tests recorded (tests call `note(key, value)` of their module's WORST dict): printed
    with every run so that the driver's record shows how far inside its bounds each case landed."""
    rows = []
    for name, mod in list(sys.modules.items()):
        if name.startswith("tests.") or name.startswith("test_"):
            w = getattr(mod, "WORST", None)
            if isinstance(w, dict):
                rows += sorted(w.items())
    # keys that start with "~" go last: facts the driver's truncated tail must keep (the batch the full-size C5 case ran)
    rows = [r for r in rows if not r[0].startswith("~")] + [r for r in rows if r[0].startswith("~")]
    if rows:
        terminalreporter.section("worst errors recorded by the parity tests")
        for key, val in rows:
            terminalreporter.write_line("%.3e  %s" % (val, key))
