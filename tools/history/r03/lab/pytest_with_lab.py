"""Run (a part of) the GPU test suite against an EXPERIMENTAL library build: pre-validation of a lab variant before it is promoted.

    python tools/lab/pytest_with_lab.py <experiment> [pytest arguments ...]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from omnivggt_official_amd import build as B, lib as L  # noqa: E402

L.LIB_PATH = os.path.join(HERE, "_build", sys.argv[1], "libomnivggt_hip.so")
assert os.path.exists(L.LIB_PATH), L.LIB_PATH
B.is_current = lambda: True
import pytest  # noqa: E402

sys.exit(pytest.main(sys.argv[2:]))
