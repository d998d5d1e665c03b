import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import gsbp_amd  # noqa: E402,F401  (first thing in the test process: the package asks the HIP runtime for the hardware queues its view
                 # pipeline needs, which only works before the first HIP call)


# DEVELOPER hook of the TEST HARNESS (never of the product, which does not look at the environment to find its library): run the
# suite against another build of the same C ABI -- tools/alt_build_test.sh builds -O2 / -O1 variants into tools/lib/ and sets this.
if os.environ.get("GWBP_TEST_LIB"):
    gsbp_amd._lib.use_library(os.environ["GWBP_TEST_LIB"], allow_profile=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
