import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import oracle
    oracle.build()


@pytest.fixture
def lib_option():
    """`lib_option(name, value)` sets a switch of libfocnerf_hip.so (include/focnerf.h foc_set_option) for the rest of the test; every
    switch touched goes back to what it was afterwards. FOC_OCC_MARCH_FORM takes its names ("two", "row", "lane", "staged"; "" = by
    burst length). The library reads its environment once, at load: setting the variable inside a test would reach nothing."""
    from focnerf_amd import _lib
    forms = {"": -1, "two": 0, "row": 1, "lane": 2, "staged": 3}
    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = _lib.get_option(name)
        _lib.set_option(name, forms[value] if isinstance(value, str) and value in forms else int(value))
    yield set_
    for name, value in saved.items():
        _lib.set_option(name, value)
