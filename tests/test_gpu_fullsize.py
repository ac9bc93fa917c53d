"""BASELINE.json's configurations at their FULL sizes on one MI355X (configs A, B, C; E's shape lives in
tests/test_gpu_parity_3d.py because the oracle needs a small volume).

Size-independent properties + a strided oracle sample, the checks of tests/fullsize/run_configs.py:
  * GPU == oracle(OC_ORDER_LANES) bit for bit on every k-th POI of the queue (FFTCC by the GPU, ICGN by the oracle),
  * split queue == whole queue bit for bit (what multi-GPU sharding relies on: a POI's result does not depend on
    which block of the queue it travels in),
  * the analytic displacement field the synthetic pair was rendered with is recovered,
  * (almost) every POI converges.
"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _configs():
    spec = importlib.util.spec_from_file_location("run_configs", os.path.join(ROOT, "tests", "fullsize", "run_configs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _check(rec, min_converged, max_err):
    assert rec["oracle_bit_exact"], rec
    assert rec["split_queue_same_bits"], rec
    assert rec["converged"] >= min_converged * rec["pois"], rec
    assert rec["median_abs_err_u"] < 0.01 and rec["max_abs_err_u"] < max_err and rec["max_abs_err_v"] < max_err, rec


def test_config_a_full_size():
    """A: 2048^2, r = 15 (FFTCC2D fused 30x30 + ICGN2D1), 100 x 100 POIs -- every POI against the oracle."""
    rec = _configs().run_2d("A", 2048, 15, 100, 1, 10000)
    _check(rec, 0.999, 0.05)
    assert rec["oracle_sample"] == 10000


def test_config_b_full_size():
    """B (the bench line): 4096^2, r = 16, 500 x 500 POIs, FFTCC2D + ICGN2D1; every 125th POI against the oracle."""
    rec = _configs().run_2d("B", 4096, 16, 500, 1, 2000)
    _check(rec, 0.999, 0.05)
    assert rec["pois"] == 250000 and rec["oracle_sample"] >= 2000


def test_config_c_full_size():
    """C: 4096^2, r = 20, ICGN2D2 (12 DoF), 316 x 316 POIs, second-order displacement field."""
    rec = _configs().run_2d("C", 4096, 20, 316, 2, 1000, so=dict(uxx=2e-6, vyy=-1e-6))
    _check(rec, 0.99, 0.1)
    assert rec["pois"] == 99856
