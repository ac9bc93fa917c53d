"""Drop-in headers vs the reference's own user code (CPU side).

* The reference's example mains compile UNMODIFIED against include/opencorr_compat (needs /root/reference: skipped on the
  GPU box, where tests/test_gpu_reference_examples.py runs the binaries built from them).
* Image2D(path) / Image3D(path) decode what the reference's fixtures are (8-bit BMP, `.bin` volumes) plus 24-bit BMP and
  binary PGM, with cv::imread(..., IMREAD_GRAYSCALE)'s grey weights; a file that is not an image fails like a failed imread.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "opencorr_amd", "lib")
REF = "/root/reference"

EXAMPLES = ["test_2d_dic_fftcc_icgn1", "test_2d_dic_fftcc_iclm1", "test_2d_dic_fftcc_nr1", "test_2d_dic_strain",
            "test_dvc_fftcc_icgn1", "test_dvc_strain", "test_dvc_gpu_icgn"]


@pytest.mark.parametrize("name", EXAMPLES)
def test_reference_example_compiles_unmodified(name):
    src = os.path.join(REF, "examples", name + ".cpp")
    if not os.path.exists(src):
        pytest.skip("reference tree not mounted")
    cmd = ["g++", "-std=c++17", "-fopenmp", "-fsyntax-only", "-w", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "include", "opencorr_compat"), src]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()[-3000:]


@pytest.fixture(scope="module")
def decoder(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("dec") / "decode_image")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "decode_image.cpp"), "-o", exe, "-L" + LIBDIR, "-lopencorr_hip",
                           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _decode(decoder, kind, path, tmp_path):
    out = str(tmp_path / "out.bin")
    rc = subprocess.call([decoder, kind, str(path), out], stderr=subprocess.DEVNULL)
    if rc != 0:
        return rc, None
    raw = open(out, "rb").read()
    dx, dy, dz = struct.unpack("<3i", raw[:12])
    return 0, np.frombuffer(raw[12:], np.float32).reshape(dz, dy, dx)


def test_image2d_decodes_bmp_and_pgm(decoder, tmp_path, golden):
    from test_gpu_reference_examples import _write_bmp8
    img = golden["ref"].astype(np.uint8)[:61, :37]   # odd width: rows are padded
    _write_bmp8(tmp_path / "a.bmp", img)
    rc, got = _decode(decoder, "2d", tmp_path / "a.bmp", tmp_path)
    assert rc == 0 and np.array_equal(got[0], img.astype(np.float32))
    # 24-bit, top-down (negative height), colour: OpenCV's fixed-point grey
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (13, 10, 3), dtype=np.uint8)
    row = (10 * 3 + 3) // 4 * 4
    body = np.zeros((13, row), np.uint8)
    body[:, :30] = rgb[:, :, ::-1].reshape(13, 30)   # B G R
    with open(tmp_path / "c.bmp", "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + body.size, 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, 10, -13, 1, 24, 0, body.size, 2835, 2835, 0, 0))
        f.write(body.tobytes())
    rc, got = _decode(decoder, "2d", tmp_path / "c.bmp", tmp_path)
    r, g, b = (rgb[:, :, k].astype(np.uint32) for k in range(3))
    want = ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.float32)
    assert rc == 0 and np.array_equal(got[0], want)
    # binary PGM with a comment line
    with open(tmp_path / "p.pgm", "wb") as f:
        f.write(b"P5\n# made by a test\n37 61\n255\n" + img.tobytes())
    rc, got = _decode(decoder, "2d", tmp_path / "p.pgm", tmp_path)
    assert rc == 0 and np.array_equal(got[0], img.astype(np.float32))
    # not an image: the reference throws std::string("Fail to load file: ...") (src/oc_image.cpp:41-44)
    (tmp_path / "x.bmp").write_bytes(b"BMnot really")
    assert _decode(decoder, "2d", tmp_path / "x.bmp", tmp_path)[0] == 4
    assert _decode(decoder, "2d", tmp_path / "missing.bmp", tmp_path)[0] == 4


def test_image2d_decodes_the_reference_fixture_like_pil(decoder, tmp_path, golden):
    path = os.path.join(REF, "examples", "2d_dic", "oht_cfrp_0.bmp")
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    rc, got = _decode(decoder, "2d", path, tmp_path)
    assert rc == 0 and np.array_equal(got[0], golden["ref"])


def test_image3d_loads_bin_volumes(decoder, tmp_path):
    vol = np.random.default_rng(5).random((5, 7, 9)).astype(np.float32)   # z, y, x
    with open(tmp_path / "v.bin", "wb") as f:
        f.write(struct.pack("<3i", 9, 7, 5))   # header: dim x, y, z (src/oc_image.cpp:93-99)
        f.write(vol.tobytes())
    rc, got = _decode(decoder, "3d", tmp_path / "v.bin", tmp_path)
    assert rc == 0 and np.array_equal(got, vol)
    assert _decode(decoder, "3d", tmp_path / "v.tif", tmp_path)[0] == 4
