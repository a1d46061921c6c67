import importlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import oracle_lib as ol  # noqa: E402

GOLD = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The HIP runtime ends the process with a bare abort() when a queue reports an error (its message is only printed at
    # log level 1 and above); round 3 lost one such abort without a word on stderr.  Errors only: no trace.
    os.environ.setdefault("AMD_LOG_LEVEL", "1")


def load_cases():
    with open(os.path.join(GOLD, "cases.json")) as f:
        return json.load(f)


def case_input(case):
    """Rebuilds the input frame of a golden case (inputs are pinned generators or
    committed data files; expected outputs come from the reference build)."""
    name = case["name"]
    if name.startswith(("fruit_tiled_", "std420_fruit_tiled_")):  # src[(y mod 254) * 253 + (x mod 253)]
        fruit = ol.read_ppm(os.path.join(GOLD, "fruit.ppm"))
        yy, xx = np.arange(case["H"]) % fruit.shape[0], np.arange(case["W"]) % fruit.shape[1]
        return np.ascontiguousarray(fruit[yy][:, xx])
    if name.startswith("fruit"):
        return ol.read_ppm(os.path.join(GOLD, "fruit.ppm"))
    if name.startswith("lcg_"):
        return ol.lcg_frame(case["W"], case["H"], case["seed"])
    z = np.load(os.path.join(GOLD, "structured_inputs.npz"))
    return z[name.rsplit("_q", 1)[0]]


@pytest.fixture(scope="session")
def jpeg():
    """The product binding (ctypes over libmi355jpeg.so)."""
    return importlib.import_module("jpeg-encoder-opencl_amd")


@pytest.fixture(scope="session")
def enc(jpeg):
    """One GPU encode context.  Fails loudly (no skip, no CPU fallback) when the HIP
    library or the device is missing."""
    e = jpeg.Encoder(0)
    yield e
    e.close()
