"""Host-side checks of bench.py that need no GPU: the algorithmic-bytes formula of SURVEY 8(d) and the keys of the
JSON line the driver parses."""
import ast
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_formula():
    import bench
    pois = np.zeros((5, 25), np.float32)
    pois[:, 17] = [3, 4, 0, 10, 2]  # iterations; a POI that never iterated (guard) only moves its record
    n2 = 33 * 33
    total, mean_iter = bench.algorithmic_bytes_icgn2d1(pois, 16, 16)
    want = 4 * (3 * n2 * 4 + 200) + (3 + 4 + 10 + 2) * n2 * 64 + 200   # B1 = 3*N2*4 + k*N2*64 + 200 per POI
    assert total == want and abs(mean_iter - 19 / 4) < 1e-12
    # SURVEY's worked number: k = 4.5 at r = 16 -> 326.9 KB per POI
    assert abs((3 * n2 * 4 + 200 + 4.5 * n2 * 64) - 326_900) < 100


def test_bench_prints_the_contract_keys():
    """The one JSON line carries every key of the driver's contract plus roofline and cpu_baseline."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    keys = {c.value for n in ast.walk(tree) if isinstance(n, ast.Dict) for c in n.keys if isinstance(c, ast.Constant)}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "workload", "roofline", "bound", "achieved", "peak", "frac",
              "traffic", "cores", "kind", "sample", "value_device_resident", "value_pcie_inclusive", "frac_algorithmic",
              "hbm_frac_by_counters", "traffic_stale"):
        assert k in keys or ('"%s"' % k) in src, k
    assert 'out["cpu_baseline"]' in src and "--no-cpu-baseline" in src


def test_algorithmic_flops_formula_and_roofline_fractions_are_bounded():
    """roofline.frac must be a fraction.  Main roof: SURVEY 8(d) bytes over the guide's L2 figure (~0.48 at the measured
    3.5 ms per 250 000-POI launch); `valu`: flops of the reference's arithmetic over the fp32 vector peak for separately
    rounded operations (~0.28)."""
    import bench
    pois = np.zeros((3, 25), np.float32)
    pois[:, 17] = [3, 0, 5]
    n2 = 33 * 33
    assert bench.algorithmic_flops_icgn2d1(pois, 16, 16) == 2 * 50 * n2 + 8 * 75 * n2
    assert abs(bench.VALU_PEAK_TFLOPS - 78.6432) < 1e-3
    per_poi = 50 * n2 + 75 * n2 * 3.105
    frac = per_poi * 250000 / 3.5e-3 / 1e12 / bench.VALU_PEAK_TFLOPS
    assert 0.2 < frac < 0.35
    bytes_per_poi = 3 * n2 * 4 + 200 + 3.105 * n2 * 64
    rate = bytes_per_poi * 250000 / 3.5e-3 / 1e9
    assert 0.3 < rate / bench.L2_PEAK_GBS < 0.6
    assert 0.45 < rate / bench.GATHER_UBENCH_GBS < 0.75 and 0.6 < rate / bench.GATHER_UBENCH_FREE_GBS < 1.0
    assert abs(bench.L1_PORT_PEAK_GBS - 39321.6) < 1e-6


def test_roofline_block_is_well_formed():
    """The roofline object is built by a plain function: exercise it with the measured figures (a stray '%' in one of
    its strings once crashed the bench at the very end of a run)."""
    import json
    import bench
    n2 = 33 * 33
    alg_bytes = (3 * n2 * 4 + 200 + 3.105 * n2 * 64) * 250000
    alg_flops = (50 * n2 + 75 * n2 * 3.105) * 250000
    r = bench.roofline_block(alg_bytes, alg_flops, 3.23, 10, {"hbm_bytes_per_launch": 2.4e9, "l2_bytes_per_launch": 5.5e10,
                                                              "source": "profiles/icgn2d1_traffic_configB.json"})
    json.dumps(r)
    assert r["bound"] == "l2" and r["unit"] == "GB/s"
    # traffic is the PMC record of the committed profiling run over the same command (HBM side and L2 side)
    assert r["traffic"] == 2.4e9 and r["traffic_l2"] == 5.5e10 and r["traffic_source"].startswith("profiles/")
    assert abs(r["l2_counter_frac"] - 5.5e10 / 3.23e-3 / 1e9 / bench.L2_PEAK_GBS) < 1e-9
    # `frac` is the COUNTER fraction of the level `bound` names (L2 request bytes / time / the guide's L2 figure); the SURVEY 8(d)
    # byte count over the same peak travels beside it as frac_algorithmic; the HBM side by counters at the block's top level
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and abs(r["frac"] - r["l2_counter_frac"]) < 1e-12
    assert abs(r["frac_algorithmic"] - alg_bytes / 3.23e-3 / 1e9 / bench.L2_PEAK_GBS) < 1e-9 and 0.45 < r["frac_algorithmic"] < 0.6
    assert abs(r["hbm_frac_by_counters"] - 2.4e9 / 3.23e-3 / 1e9 / bench.HBM_PEAK_GBS) < 1e-12 and r["traffic_stale"] is None
    assert 0.5 < r["gather_ubench"]["frac"] < 0.7 and 0.25 < r["valu"]["frac"] < 0.34
    assert abs(r["l1_port"]["frac"] - r["achieved_algorithmic"] / bench.L1_PORT_PEAK_GBS) < 1e-12
    assert "2.2x" in r["why_not_hbm"]
    z = bench.roofline_block(alg_bytes, alg_flops, 0.0, 0, None)   # nothing timed: no division by zero
    assert z["achieved"] == 0.0 and z["frac"] == 0.0 and z["traffic"] is None
    # a PMC record that belongs to other kernel sources: no counter figure is reported, the reason is, frac falls back to the
    # algorithmic bytes and says so
    st = bench.roofline_block(alg_bytes, alg_flops, 3.23, 10, {"stale": "collected on other kernel sources", "source": "profiles/x.json"})
    assert st["traffic"] is None and st["traffic_l2"] is None and st["hbm_frac_by_counters"] is None
    assert st["traffic_stale"] == "collected on other kernel sources" and st["frac"] == st["frac_algorithmic"]
    assert "no PMC record" in st["achieved_source"]


def test_pmc_records_are_bound_to_the_kernel_sources(tmp_path):
    """tools/pmc_traffic.py stores a fingerprint of the kernel family's sources + compiler flags in every PMC record; bench.py
    recomputes it and reports `traffic: null` plus the reason when they differ (a kernel edit must not keep old counters)."""
    import json
    import bench
    from opencorr_amd import build as hip_build
    fp = hip_build.kernel_fingerprint("icgn2d_kernel")
    assert fp == hip_build.kernel_fingerprint("icgn2d_kernel") and fp != hip_build.kernel_fingerprint("icgn3d1")
    rec = {"hbm_bytes_per_launch": 2.5e9, "l2_bytes_per_launch": 3.9e10, "collected": "2026-10-01", "source_fingerprint": fp,
           "per_kernel": {"icgn2d_kernel": {"hbm_bytes_per_launch": 2.5e9, "l2_bytes_per_launch": 3.9e10, "source_fingerprint": fp}}}
    good = tmp_path / "good.json"
    good.write_text(json.dumps(rec))
    got = bench.pmc_profile_from(str(good))
    assert got["hbm_bytes_per_launch"] == 2.5e9 and "stale" not in got
    ok, why = bench.traffic_record_is_current(rec, "icgn2d_kernel")
    assert ok and why is None
    for broken in (dict(rec, source_fingerprint="0123456789abcdef"), {k: v for k, v in rec.items() if k != "source_fingerprint"}):
        bad = tmp_path / "bad.json"
        bad.write_text(json.dumps(broken))
        got = bench.pmc_profile_from(str(bad))
        assert "stale" in got and got.get("hbm_bytes_per_launch") is None and "gpu_profiles.sh" in got["stale"]
        blk = bench.roofline_block(1e10, 1e9, 3.2, 5, got)
        assert blk["traffic"] is None and blk["traffic_stale"] == got["stale"]
    # every committed record the bench line reads is either current or reported as stale -- never silently used
    for path in list(bench.TRAFFIC_BY_CONFIG.values()) + list(bench.TRAFFIC_BY_CONFIG_FMA.values()):
        if os.path.exists(path):
            r = json.load(open(path))
            for name, k in (r.get("per_kernel") or {}).items():
                ok, why = bench.traffic_record_is_current(k, name)
                assert ok or why


def test_secondary_roofline_blocks():
    """One block per dominant kernel of the other BASELINE configs, each recomputable from its own fields."""
    import json
    import bench
    n3 = 33 ** 3
    alg = (4 * n3 * 4 + 248 + 2.906 * n3 * 256) * 50653   # 4*N3*4 + k*N3*256 + 248 B per POI
    b = bench.secondary_block("icgn3d1_kernel (ICGN3D1)", "E", alg, 80.7, 3, "lds", bench.LDS_READ2_PEAK_GBS, "note")
    json.dumps(b)
    assert abs(bench.LDS_READ2_PEAK_GBS - 78643.2) < 1e-6
    assert abs(b["achieved"] - alg / 80.7e-3 / 1e9) < 1e-6 and abs(b["frac"] - b["achieved"] / b["peak"]) < 1e-12
    assert 0.18 < b["frac"] < 0.25    # the round-2 kernel: 0.21 of the ds_read2_b32 rate
    z = bench.secondary_block("k", "c", 1.0, 0.0, 0, "hbm", bench.HBM_PEAK_GBS, "")
    assert z["achieved"] == 0.0


def test_the_default_run_has_no_hidden_warmup():
    import bench
    import sys
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert a.settle == 0 and a.gpus == 1


def test_weak_layout_keeps_the_poi_pitch_at_every_n():
    """Weak scaling: the image is a x b tiles of 4096^2 px with 500 x 500 POIs each, so the POI pitch (how much
    neighbouring subsets overlap, i.e. the kernel's cache behaviour per POI) is N = 1's at every N -- also at N = 8, which
    round 3 ran on a denser grid -- and the image stays inside the 2^28-pixel limit of the 32-bit LUT plane offsets."""
    import bench
    base = bench.weak_layout(1)
    assert base == (4096, 4096, 500, 500)
    for n in range(1, 17):
        h, w, nx, ny = bench.weak_layout(n)
        assert nx * ny == n * 250000 and h * w == n * 4096 * 4096 <= 2 ** 28
        assert (w / nx, h / ny) == (4096 / 500, 4096 / 500)
        assert h <= w < 2 ** 22
    assert bench.weak_layout(8) == (8192, 16384, 2000, 1000)
    assert bench.weak_layout(2, 512, 40) == (512, 1024, 80, 40)
    with pytest.raises(SystemExit):
        bench.weak_layout(17)
