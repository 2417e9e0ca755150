"""The optional GEMM kernel variants (A/B switches read once per process from the environment) stay correct: each is
exercised by tools/check_gemm_variant.py in its own short subprocess, one after the other."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS_LIB = os.path.join(ROOT, "tools", "diag", "libleaf_hip_variants.so")   # make -C leaf_amd/csrc variants (__graft_entry__.build())

VARIANTS = [
    {"LEAF_GEMM256H": "0"},                          # two-stage 256^2 + register-staged kernels instead of the half-stage ring
    {"LEAF_GEMM64_DEEP": "1", "LEAF_GEMM64_MI": "2"},  # 6-slot small-launch ring
    {"LEAF_GEMM64": "0", "LEAF_GEMM_BM64": "1"},     # register-staged 64-row tiles
    {"LEAF_GEMM_PP": "1", "LEAF_GEMM_PP_MIN_TILES": "32"},   # 128 x 256 tiles, two workgroups per CU (variants/gemm128pp.hip: diagnostic build only)
]


def test_pingpong_gemm_rows_have_the_same_bits_as_every_other_kernel():
    """variants/gemm128pp.hip (the round-3 two-workgroups-per-CU experiment, LEAF_GEMM_PP=1) through the bit-exactness tests of the GEMM
    family: big launch (ping-pong kernel) against chunked launches (small-launch kernels), all epilogues incl. the LN-folded ones."""
    full = dict(os.environ, LEAF_GEMM_PP="1", LEAF_GEMM_PP_MIN_TILES="64", LEAF_HIP_LIB=VARIANTS_LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_kernels.py"), "-x", "-q", "-k",
                        "rows_do_not_depend or lnfold"], env=full, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_gemm_variant(env):
    full = dict(os.environ)
    full.update(env)
    if "LEAF_GEMM_PP" in env:
        full["LEAF_HIP_LIB"] = VARIANTS_LIB      # the ping-pong kernel exists only in the diagnostic build
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_gemm_variant.py")], env=full, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_four_wave_gemm_experiment_has_the_same_bits_as_the_shipped_kernel():
    """variants/gemm256w4.hip (round 5: the 256 x 256 tile as FOUR waves of 128 x 128, the wave shape of the vendor's hipBLASLt kernel;
    measured no faster, profiles/r05_w4_experiment.txt, kept as a documented experiment in the diagnostic build): same bits as the
    eight-wave kernel on every shape it takes (SHA-1 of the whole output, tools/w4_bench.py), incl. a ragged last M tile."""
    out = {}
    for w4 in ("0", "1"):
        env = dict(os.environ, LEAF_HIP_LIB=VARIANTS_LIB, LEAF_GEMM_W4=w4, ROWS="33001")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "w4_bench.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out[w4] = [l.split("sha1")[1].strip() for l in r.stdout.splitlines() if "sha1" in l]
        assert all(float(l.split("max|err| vs torch (512 rows)")[1].split()[0]) < 0.02 for l in r.stdout.splitlines() if "sha1" in l)
    assert len(out["0"]) == 4 and out["0"] == out["1"], out
