import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """The reference's own OHT-CFRP example (images + result CSVs), see tests/golden/make_golden.py."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "oht_cfrp_r16.npz"))
    return dict(ref=g["ref"].astype(np.float32), tar=g["tar"].astype(np.float32), table=g["table"],
                deformation=g["deformation"], rx=int(g["params"][0]), ry=int(g["params"][1]),
                stop=float(g["params"][2]), conv=float(g["conv"]))


@pytest.fixture(scope="session")
def golden_icgn2():
    """x y u v u0 v0 zncc iteration convergence of the reference authors' CUDA ICGN2D2 run on the OHT pair."""
    return np.load(os.path.join(ROOT, "tests", "golden", "oht_cfrp_sift_icgn2_gpu_r16.npz"))["table"]


@pytest.fixture(scope="session")
def golden_nr1():
    """x y u v u0 v0 zncc iteration convergence of the reference's FFTCC2D -> NR2D1 example on the OHT pair."""
    return np.load(os.path.join(ROOT, "tests", "golden", "oht_cfrp_fftcc_nr1_r16.npz"))["table"]


@pytest.fixture(scope="session")
def golden_strain():
    """x y u v zncc exx eyy exy of the reference's Strain example on the OHT table (radius 20, 5 neighbours)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "oht_cfrp_strain_r20.npz"))
    return dict(table=z["table"], radius=float(z["params"][0]), neighbors=int(z["params"][1]),
                zncc_threshold=float(z["params"][2]), approximation=int(z["params"][3]))


@pytest.fixture(scope="session")
def speckle_small():
    """320 x 300 synthetic speckle pair with the SURVEY 8(d) displacement field."""
    from opencorr_amd import synth
    ref, tar = synth.speckle_pair_2d(300, 320, seed=20260925)
    return ref, tar
