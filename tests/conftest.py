import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A hung kernel must fail ONE test, not hold the GPU box until the driver's limit: every GPU test gets a wall-clock bound
    (pytest-timeout is in the image; the full-size depth-24 cases get more)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if "gpu" in it.keywords and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(1500 if "full_size" in it.name else 420))


@pytest.fixture(scope="session")
def fixture_tree():
    return dict(np.load(os.path.join(GOLDEN, "tdm_tree.npz")))


@pytest.fixture(scope="session")
def fixture_w32():
    return np.load(os.path.join(GOLDEN, "din_f32.npy"))


@pytest.fixture(scope="session")
def fixture_w64():
    return np.load(os.path.join(GOLDEN, "din_f64.npy"))


@pytest.fixture(scope="session")
def fixture_otm_mapping():
    return np.load(os.path.join(GOLDEN, "otm_mapping.npy"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def oracle_tree(oracle, fixture_tree):
    t = fixture_tree
    return oracle.TdmTree(t["codes"], t["ids"], t["is_leaf"], t["leaf_ids"], t["leaf_codes"], t["max_level"])


@pytest.fixture(scope="session")
def oracle_din32(oracle, fixture_w32):
    return oracle.Din(fixture_w32, 16, 10, 8191)


@pytest.fixture(scope="session")
def oracle_din64(oracle, fixture_w64):
    return oracle.Din(fixture_w64, 16, 10, 8191)


@pytest.fixture(scope="session")
def engine_fixture(fixture_tree, fixture_w32):
    """GPU engine loaded with the reference's bundled tree + trained E=16 DIN (f32)."""
    from dismember_amd import Engine
    eng = Engine(0)
    t = fixture_tree
    eng.load_tree(t["codes"], t["ids"], t["is_leaf"], int(t["max_level"]))
    eng.load_id_maps(t["leaf_ids"], t["leaf_codes"])
    eng.load_weights_din(fixture_w32, 16, 8191)
    yield eng
    eng.close()
