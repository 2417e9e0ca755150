"""The optional GEMM kernel variants (A/B switches read once per process from the environment) stay correct: each is
exercised by tools/check_gemm_variant.py in its own short subprocess, one after the other."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    {"LEAF_GEMM256H": "0"},                          # two-stage 256^2 + register-staged kernels instead of the half-stage ring
    {"LEAF_GEMM64_DEEP": "1", "LEAF_GEMM64_MI": "2"},  # 6-slot small-launch ring
    {"LEAF_GEMM64": "0", "LEAF_GEMM_BM64": "1"},     # register-staged 64-row tiles
]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_gemm_variant(env):
    full = dict(os.environ)
    full.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_gemm_variant.py")], env=full, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
