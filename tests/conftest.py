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


# ---- measured margins: tests that hold the engine to the oracle record how close they actually came (max over cases), so
# that the bars written in the tests can be set from measurements (profiles/r03_test_margins.json is such a dump).
MARGINS = {}


def record_margin(name: str, value: float):
    MARGINS[name] = max(MARGINS.get(name, 0.0), float(value))


def pytest_sessionfinish(session, exitstatus):
    out = os.path.join(REPO, "gpurun_out")
    if MARGINS and os.path.isdir(out):
        import json
        with open(os.path.join(out, "test_margins.json"), "w") as f:
            json.dump({k: round(v, 6) for k, v in sorted(MARGINS.items())}, f, indent=1)
