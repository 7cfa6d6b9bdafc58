"""pytest configuration: markers, library fixtures.

`oracle` (CPU restatement, test infrastructure) is built on demand with oracle/Makefile.
`hip` is the product library; `-m gpu` tests require it AND a GPU and fail loudly otherwise.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lio-mapping_amd"))
sys.path.insert(0, ROOT)

ORACLE_SO = os.path.join(ROOT, "oracle", "liblio_oracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from lio_amd import capi

    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return capi.LioLib(ORACLE_SO)


@pytest.fixture(scope="session")
def hip():
    from lio_amd import capi

    return capi.load_hip()  # raises when missing: no CPU fallback
