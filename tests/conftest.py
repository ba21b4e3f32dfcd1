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


@pytest.fixture(scope="session")
def mo():
    import mpc_oracle
    return mpc_oracle


@pytest.fixture(scope="session")
def co():
    import c_oracle
    c_oracle.build()
    return c_oracle


@pytest.fixture(scope="session")
def pkg():
    import almpc_loader
    return almpc_loader.load_package()


@pytest.fixture(scope="session")
def capi(pkg):
    if not os.path.exists(pkg._capi.LIB_PATH):  # fresh checkout (the .so is not in git): build it once, as the driver's build() does
        import __graft_entry__
        __graft_entry__.build()
    pkg._capi.load()
    return pkg._capi


@pytest.fixture(scope="session")
def qtp_ab(mo):
    with open(os.path.join(GOLDEN, "linear_regressor_train_result.jls"), "rb") as f:
        return mo.decode_linear_regressor_fixture(f.read())
