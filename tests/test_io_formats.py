"""The reference's table / volume file formats (opencorr_amd/io.py)."""
import os

import numpy as np

from opencorr_amd import io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_table2d_round_trip_and_reference_layout(tmp_path):
    rng = np.random.default_rng(1)
    pois = rng.uniform(-50, 50, (7, 25)).astype(np.float32)
    p = tmp_path / "t.csv"
    io.save_table2d(p, pois)
    lines = open(p).read().splitlines()
    assert lines[0] == "x,y,u,v,u0,v0,ZNCC,iteration,convergence,feature,exx,eyy,exy,subset_rx,subset_ry,"
    assert lines[1].endswith(",") and lines[1].count(",") == 15
    assert lines[1].split(",")[0] == "%.8f" % pois[0, 0]          # fixed, 8 decimals (src/oc_io.cpp:320-322)
    back = io.load_table2d(p)
    cols = [off for _, off in io.TABLE2D]
    assert np.allclose(back[:, cols], pois[:, cols], atol=1e-5)
    untouched = [c for c in range(25) if c not in cols]
    assert (back[:, untouched] == 0).all()


def test_golden_csv_of_the_reference_parses(golden, tmp_path):
    """A file written with the golden table's values reads back into the POI2D layout the tests use."""
    tab = golden["table"][:50]
    pois = np.zeros((50, 25), np.float32)
    pois[:, [0, 1, 2, 8, 14, 15, 16, 17, 18]] = tab
    p = tmp_path / "g.csv"
    # the reference's older files stop after "convergence": emulate by truncating the columns
    with open(p, "w") as f:
        f.write("x,y,u,v,u0,v0,ZNCC,iteration,convergence,\n")
        for r in tab:
            f.write(",".join("%.8f" % v for v in r) + ",\n")
    back = io.load_table2d(p)
    assert np.allclose(back, pois, atol=1e-6)


def test_deformation_and_3d_tables(tmp_path):
    rng = np.random.default_rng(2)
    p2 = rng.uniform(-1, 1, (4, 25)).astype(np.float32)
    io.save_deformation_table2d(tmp_path / "d.csv", p2)
    head = open(tmp_path / "d.csv").readline().strip()
    assert head == "x,y,u,ux,uy,uxx,uxy,uyy,v,vx,vy,vxx,vxy,vyy,subset_rx,subset_ry,"
    p3 = rng.uniform(-1, 1, (5, 31)).astype(np.float32)
    io.save_table3d(tmp_path / "t3.csv", p3)
    back = io.load_table3d(tmp_path / "t3.csv")
    assert np.allclose(back, p3, atol=1e-6)   # every POI3D field is a column of the 3D table


def test_bin_volume_round_trip(tmp_path):
    vol = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    io.save_bin_volume(tmp_path / "v.bin", vol)
    raw = np.fromfile(tmp_path / "v.bin", dtype=np.int32, count=3)
    assert raw.tolist() == [4, 3, 2]           # dim_x, dim_y, dim_z (src/oc_image.cpp:96-101)
    assert np.array_equal(io.load_bin_volume(tmp_path / "v.bin"), vol)


def test_cpp_io_header_exchanges_files_with_the_python_twin(tmp_path):
    """include/opencorr_compat/oc_io.h (IO2D / IO3D with the reference's names) reads what io.py writes and writes
    what io.py reads: same columns, same 8-decimal fixed notation."""
    import subprocess
    rng = np.random.default_rng(4)
    p2 = np.zeros((6, 25), np.float32)
    p2[:, :2] = [[1, 2], [3, 4], [5, 6], [7, 1], [9, 8], [11, 3]]
    p2[:, 2:] = rng.uniform(-3, 3, (6, 23)).astype(np.float32)
    p3 = rng.uniform(-3, 3, (5, 31)).astype(np.float32)
    io.save_table2d(tmp_path / "in2.csv", p2)
    io.save_table3d(tmp_path / "in3.csv", p3)
    exe = str(tmp_path / "io_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "io_driver.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "opencorr_amd", "lib"), "-lopencorr_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "opencorr_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    out = [str(tmp_path / n) for n in ("out2.csv", "outd.csv", "in3.csv", "out3.csv", "map.csv")]
    subprocess.check_call([exe, str(tmp_path / "in2.csv"), out[0], out[1], out[2], out[3], out[4]])
    # the C++ writer reproduces the Python writer's file byte for byte (both print what was read back from 8 decimals)
    io.save_table2d(tmp_path / "ref2.csv", io.load_table2d(tmp_path / "in2.csv"))
    assert open(out[0]).read() == open(tmp_path / "ref2.csv").read()
    io.save_table3d(tmp_path / "ref3.csv", io.load_table3d(tmp_path / "in3.csv"))
    assert open(out[3]).read() == open(tmp_path / "ref3.csv").read()
    io.save_deformation_table2d(tmp_path / "refd.csv", io.load_table2d(tmp_path / "in2.csv"))
    # the deformation table carries fields the result table does not (ux, uy, ...): they were zero after the load
    assert open(out[1]).read() == open(tmp_path / "refd.csv").read()
    m = np.loadtxt(out[4], delimiter=",", usecols=range(12))
    assert m.shape == (9, 12)
    back = io.load_table2d(tmp_path / "in2.csv")
    for row in back:
        assert abs(m[int(row[1]), int(row[0])] - row[21]) < 1e-6
    assert np.count_nonzero(m) == 6


def test_ragged_rows_keep_their_own_width(tmp_path):
    """A truncated line must cost only itself its missing columns (the C++ loader works row by row too)."""
    from opencorr_amd import io
    pois = np.zeros((3, 25), np.float32)
    pois[:, 0] = [10, 20, 30]
    pois[:, 1] = [11, 21, 31]
    pois[:, 2] = [0.5, 0.25, 0.125]     # u
    pois[:, 16] = [0.9, 0.8, 0.7]       # zncc
    path = tmp_path / "t.csv"
    io.save_table2d(str(path), pois)
    lines = open(path).read().splitlines()
    lines[2] = ",".join(lines[2].split(",")[:3]) + ","   # second POI: only x, y, u survive
    lines.append("42,")                                      # a row without a position: skipped
    open(path, "w").write("\n".join(lines) + "\n")
    got = io.load_table2d(str(path))
    assert got.shape == (3, 25)
    assert np.array_equal(got[[0, 2]], pois[[0, 2]])
    assert got[1, 0] == 20 and got[1, 1] == 21 and got[1, 2] == 0.25 and got[1, 16] == 0
