import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def pytest_sessionstart(session):
    """Initialise torch's HIP runtime BEFORE liblnsfaid.so is loaded.  torch ships its own libamdhip64; when
    the system one (pulled in by liblnsfaid.so) is mapped first, torch afterwards reports "No HIP GPUs".
    Tests use torch only as a device-memory allocator for the device-pointer entry points."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def abi():
    import oracle_abi
    return oracle_abi.pyabi


@pytest.fixture(scope="session")
def lib(abi):
    return abi.load()


@pytest.fixture(scope="session")
def code50(abi, lib):
    return abi.Code50GPON(lib)


@pytest.fixture(scope="session")
def encoder(code50):
    import gf2_encoder
    return gf2_encoder.Encoder(code50)
