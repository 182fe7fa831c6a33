import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def locked_model():
    from robogym_amd.envs.dactyl.locked import load_locked_model
    from robogym_amd.mujoco.kernel_tables import derive_kernel_tables

    m = load_locked_model()
    derive_kernel_tables(m)
    return m


@pytest.fixture(scope="session")
def emul_lib():
    """The HIP kernel source compiled for the host (fiber-emulated wavefront). Test harness only."""
    import subprocess

    from robogym_amd import _native

    d = os.path.join(ROOT, "tests", "emul")
    subprocess.check_call(["make", "-C", d, "-s"])
    return _native.bind(os.path.join(d, "librgstep_emul.so"))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import rg_oracle

    rg_oracle.build()
    return rg_oracle


@pytest.fixture
def kernel_variant(oracle_lib):
    """The oracle configured to the HIP kernel's two documented deviations from the MuJoCo restatement (portal-plane
    MPR depth, box-box through MPR): used by the tests that check "the kernel computes what it says" at fp32
    tolerance.  The size of the deviation itself is measured against the DEFAULT oracle by the `*_deviation_*` tests."""
    oracle_lib.set_kernel_variant(True)
    yield
    oracle_lib.set_kernel_variant(False)
