"""Fixture exchange in the reference's own file formats.

Host-side only -- nothing here touches the GPU.  These are the formats OpenCorr's examples
write and its golden vectors are stored in, so results of the HIP engines can be diffed
against files produced by the reference and vice versa:

* ``IO2D::saveTable2D / loadTable2D``            src/oc_io.cpp:249-345
  ``x,y,u,v,u0,v0,ZNCC,iteration,convergence,feature,exx,eyy,exy,subset_rx,subset_ry,``
* ``IO2D::saveDeformationTable2D``               src/oc_io.cpp:347-392
  ``x,y,u,ux,uy,uxx,uxy,uyy,v,vx,vy,vxx,vxy,vyy,subset_rx,subset_ry,``
* ``IO3D::saveTable3D / loadTable3D``            src/oc_io.cpp:1004-1089
* ``Image3D::loadBin``                           src/oc_image.cpp:76-110 -- ``int32[3]`` header
  (dim_x, dim_y, dim_z) followed by dim_z*dim_y*dim_x float32, x fastest.

Numbers are written like the reference does: fixed notation, 8 decimals, a trailing delimiter
before the newline (src/oc_io.cpp:320-322).  Older files of the reference lack the strain /
subset-radius columns (e.g. examples/2d_dic/oht_cfrp_4_fftcc_icgn1_r16.csv); the loaders accept
any prefix of the column list.
"""
import numpy as np

# column -> float offset in the POI2D record (src/oc_poi.h:102-136)
TABLE2D = [("x", 0), ("y", 1), ("u", 2), ("v", 8), ("u0", 14), ("v0", 15), ("ZNCC", 16), ("iteration", 17),
           ("convergence", 18), ("feature", 19), ("exx", 20), ("eyy", 21), ("exy", 22), ("subset_rx", 23),
           ("subset_ry", 24)]
DEFORMATION2D = [("x", 0), ("y", 1)] + [(n, 2 + i) for i, n in enumerate(
    ["u", "ux", "uy", "uxx", "uxy", "uyy", "v", "vx", "vy", "vxx", "vxy", "vyy"])] + [("subset_rx", 23), ("subset_ry", 24)]
# POI3D record (src/oc_poi.h:187-222): x y z | u ux uy uz v vx vy vz w wx wy wz | u0 v0 w0 zncc iter conv feature | e[6] | r[3]
TABLE3D = [("x", 0), ("y", 1), ("z", 2), ("u", 3), ("v", 7), ("w", 11), ("u0", 15), ("v0", 16), ("w0", 17), ("ZNCC", 18),
           ("iteration", 19), ("convergence", 20), ("feature", 21), ("ux", 4), ("uy", 5), ("uz", 6), ("vx", 8), ("vy", 9),
           ("vz", 10), ("wx", 12), ("wy", 13), ("wz", 14), ("exx", 22), ("eyy", 23), ("ezz", 24), ("exy", 25),
           ("eyz", 26), ("ezx", 27), ("subset_rx", 28), ("subset_ry", 29), ("subset_rz", 30)]


def _save(path, pois, columns, delimiter):
    pois = np.asarray(pois, dtype=np.float32)
    with open(path, "w") as f:
        f.write(delimiter.join(name for name, _ in columns) + delimiter + "\n")
        idx = [off for _, off in columns]
        for row in pois[:, idx]:
            f.write(delimiter.join("%.8f" % v for v in row) + delimiter + "\n")


def _load(path, columns, floats, delimiter):
    with open(path) as f:
        f.readline()  # header (the reference skips it without looking at it, src/oc_io.cpp:257-258)
        rows = [[float(tok) for tok in line.strip().split(delimiter) if tok != ""] for line in f if line.strip()]
    # every row fills the fields it has (a short or truncated line leaves ITS remaining fields at 0, like the C++ twin
    # include/opencorr_compat/oc_io.h does; it must not cost the other rows their columns); rows without a position
    # (fewer than 2 / 3 values) are skipped there as well
    need = 2 if floats == 25 else 3
    rows = [r for r in rows if len(r) >= need]
    pois = np.zeros((len(rows), floats), dtype=np.float32)
    for i, r in enumerate(rows):
        for value, (_, off) in zip(r, columns):
            pois[i, off] = value
    return pois


def save_table2d(path, pois, delimiter=","):
    """IO2D::saveTable2D: one row per POI2D record (n x 25 float32)."""
    _save(path, pois, TABLE2D, delimiter)


def load_table2d(path, delimiter=","):
    """IO2D::loadTable2D -> (n, 25) float32 POI2D records (fields absent from the file stay 0)."""
    return _load(path, TABLE2D, 25, delimiter)


def save_deformation_table2d(path, pois, delimiter=","):
    """IO2D::saveDeformationTable2D: the full 12-parameter deformation vector per POI."""
    _save(path, pois, DEFORMATION2D, delimiter)


def save_table3d(path, pois, delimiter=","):
    """IO3D::saveTable3D: one row per POI3D record (n x 31 float32)."""
    _save(path, pois, TABLE3D, delimiter)


def load_table3d(path, delimiter=","):
    return _load(path, TABLE3D, 31, delimiter)


def save_bin_volume(path, vol):
    """Image3D .bin: int32 dim_x, dim_y, dim_z then the voxels, x fastest (vol is indexed [z, y, x])."""
    vol = np.ascontiguousarray(vol, dtype=np.float32)
    dz, dy, dx = vol.shape
    with open(path, "wb") as f:
        np.asarray([dx, dy, dz], dtype=np.int32).tofile(f)
        vol.tofile(f)


def load_bin_volume(path):
    with open(path, "rb") as f:
        dx, dy, dz = np.fromfile(f, dtype=np.int32, count=3)
        vol = np.fromfile(f, dtype=np.float32, count=int(dx) * int(dy) * int(dz))
    return vol.reshape(int(dz), int(dy), int(dx))
