import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def lowres_assets():
    """(grid_info, input_mean, input_max, input_min, output_scale) AssetSets from the committed bundles."""
    from climsim_amd.assets import load_grid_info, load_npz_assets
    grid = load_grid_info(os.path.join(GOLDEN, "grid_lowres.npz"))
    sets = [load_npz_assets(os.path.join(GOLDEN, "norm_lowres.npz"), k)
            for k in ("input_mean", "input_max", "input_min", "output_scale")]
    return (grid, *sets)
