"""Builds tests/golden/oht_cfrp_r16.npz from the reference's own example data.

Run in the build container (needs /root/reference, which does not exist on the
GPU box):  python tests/golden/make_golden.py

Inputs (all under /root/reference/examples/2d_dic/):
  oht_cfrp_0.bmp, oht_cfrp_4.bmp           8-bit gray, 280 x 900 (cv::IMREAD_GRAYSCALE,
                                            src/oc_image.cpp:39)
  oht_cfrp_4_fftcc_icgn1_r16.csv           x,y,u,v,u0,v0,ZNCC,iteration,convergence,...
  oht_cfrp_4_fftcc_icgn1_r16_deformation.csv   x,y,u,ux,uy,v,vx,vy
produced by examples/test_2d_dic_fftcc_icgn1.cpp (r=16, conv 1e-3, stop 10,
POI grid 100 x 300, step 2, origin (30,30)).  Only data is stored -- no
reference source code.

A second, weaker anchor for the second-order engine:
  oht_cfrp_4_sift_icgn2(gpu)_r16.csv        x,y,u,v,u0,v0,ZNCC,iteration,convergence,feature
written by the reference authors' CUDA ICGN2D2 (examples/test_2d_dic_gpu_icgn.cpp) from SIFT +
FeatureAffine initial guesses.  The CSV keeps only the translation part (u0, v0) of those
guesses, so a re-run starts from a slightly different point; IC-GN is path independent enough
that the converged u, v, ZNCC still agree to ~1e-5 px / 1e-7 (see tests/test_oracle_golden.py).
Stored as oht_cfrp_sift_icgn2_gpu_r16.npz.

And the golden vectors of the Newton-Raphson engine:
  oht_cfrp_4_fftcc_nr1_r16.csv              x,y,u,v,u0,v0,ZNCC,iteration,convergence,feature,exx,eyy,exy
produced by examples/test_2d_dic_fftcc_nr1.cpp (FFTCC2D -> NR2D1, r=16, conv 1e-3, stop 10, same grid).
Stored as oht_cfrp_fftcc_nr1_r16.npz (first nine columns).

And the golden vectors of Strain: the last three columns (exx, eyy, exy) of
  oht_cfrp_4_fftcc_icgn1_r16.csv
were added to that table by examples/test_2d_dic_strain.cpp (Strain(20, 5), ZNCC threshold 0.9, Cauchy strain) from
the table's own u, v, ZNCC.  Stored as oht_cfrp_strain_r20.npz: x y u v zncc exx eyy exy in float64 (the CSV's 8
decimals) -- the table stores strains for every POI; the current source skips POIs below the ZNCC threshold
(src/oc_strain.cpp:241), so the tests compare POIs with ZNCC >= 0.9 only.
"""
import os

import numpy as np
from PIL import Image

REF = "/root/reference/examples/2d_dic"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oht_cfrp_r16.npz")


def main():
    ref = np.asarray(Image.open(os.path.join(REF, "oht_cfrp_0.bmp")))
    tar = np.asarray(Image.open(os.path.join(REF, "oht_cfrp_4.bmp")))
    assert ref.dtype == np.uint8 and ref.shape == (900, 280) and tar.shape == ref.shape
    table = np.genfromtxt(os.path.join(REF, "oht_cfrp_4_fftcc_icgn1_r16.csv"), delimiter=",", skip_header=1,
                          usecols=range(9))
    deform = np.genfromtxt(os.path.join(REF, "oht_cfrp_4_fftcc_icgn1_r16_deformation.csv"), delimiter=",",
                           skip_header=1, usecols=range(8))
    assert table.shape == (30000, 9) and deform.shape == (30000, 8)
    assert np.array_equal(table[:, :2], deform[:, :2])
    np.savez_compressed(
        OUT,
        ref=ref, tar=tar,
        # x y u v u0 v0 zncc iteration convergence
        table=table.astype(np.float32),
        # x y u ux uy v vx vy
        deformation=deform.astype(np.float32),
        params=np.array([16, 16, 10], dtype=np.int32), conv=np.float32(0.001))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    t2 = np.genfromtxt(os.path.join(REF, "oht_cfrp_4_sift_icgn2(gpu)_r16.csv"), delimiter=",", skip_header=1,
                       usecols=range(9))
    assert t2.shape == (30000, 9) and np.array_equal(t2[:, :2], table[:, :2])
    out2 = os.path.join(os.path.dirname(OUT), "oht_cfrp_sift_icgn2_gpu_r16.npz")
    # x y u v u0 v0 zncc iteration convergence
    np.savez_compressed(out2, table=t2.astype(np.float32))
    print("wrote", out2, os.path.getsize(out2), "bytes")
    t3 = np.genfromtxt(os.path.join(REF, "oht_cfrp_4_fftcc_nr1_r16.csv"), delimiter=",", skip_header=1, usecols=range(9))
    assert t3.shape == (30000, 9) and np.array_equal(t3[:, :2], table[:, :2])
    out3 = os.path.join(os.path.dirname(OUT), "oht_cfrp_fftcc_nr1_r16.npz")
    np.savez_compressed(out3, table=t3.astype(np.float32))
    print("wrote", out3, os.path.getsize(out3), "bytes")
    t4 = np.genfromtxt(os.path.join(REF, "oht_cfrp_4_fftcc_icgn1_r16.csv"), delimiter=",", skip_header=1, usecols=range(13))
    out4 = os.path.join(os.path.dirname(OUT), "oht_cfrp_strain_r20.npz")
    np.savez_compressed(out4, table=t4[:, [0, 1, 2, 3, 6, 10, 11, 12]],
                        params=np.array([20.0, 5.0, 0.9, 1.0]))  # radius, min neighbours, ZNCC threshold, approximation
    print("wrote", out4, os.path.getsize(out4), "bytes")


if __name__ == "__main__":
    main()
