"""pytest configuration: markers and import paths.

* ``gpu`` marks tests that need a real MI355X (run with ``-m gpu`` on the GPU box).
* the product package directory ``iea-gan_amd/`` is put on ``sys.path`` so that the reference's
  top-level module names (``model``, ``layers``, ``RRM``, ``loss``, ``diff_aug``, ``cr_diff_aug``,
  ``train_fns``, ``utils``) resolve to the MI355X implementation -- exactly how a user of the
  reference switches over.
* ``oracle/`` is importable from tests only (it is the checker, never the product).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "iea-gan_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP kernels execute)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ref_cfg():
    """The reference's shipped hyper-parameters (config.json), restated as data."""
    import json
    from defaults import default_config
    return default_config()
