"""jpegamd_main, the GPU leg of the reference's main.cpp harness, and
jpegamd_quality_tables (main.cpp:51-80).  The CPU tests stop before any device call."""
import ctypes
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "jpeg_amd", "jpegamd_main")


def _run(args, stdin, cwd):
    return subprocess.run([EXE, *args], input=stdin, cwd=cwd, capture_output=True, text=True, timeout=110)


def _write_ppm(path, bgr):
    with open(path, "wb") as f:
        f.write(b"P6\n# written by the test\n%d %d\n255\n" % (bgr.shape[1], bgr.shape[0]))
        f.write(np.ascontiguousarray(bgr[:, :, ::-1]).tobytes())


def _read_ppm(path):
    raw = open(path, "rb").read()
    m = re.match(rb"P6\s+(\d+)\s+(\d+)\s+255\s", raw)
    cols, rows = int(m.group(1)), int(m.group(2))
    return np.frombuffer(raw[m.end():], np.uint8).reshape(rows, cols, 3)[:, :, ::-1]


def test_quality_tables_match_main_cpp_scaling():
    import jpeg_amd
    from jpeg_amd import _lib
    lib = _lib.load()
    for q in range(1, 101):
        a, b = np.zeros(64, np.int32), np.zeros(64, np.int32)
        assert lib.jpegamd_quality_tables(q, a.ctypes.data, b.ctypes.data) == 0
        qy, qc = jpeg_amd.quality_tables(q)
        assert np.array_equal(a, qy) and np.array_equal(b, qc)
    # QF=50 is the unscaled Annex-K pair; QF=100 clamps to 1
    assert a.max() == 1 and b.max() == 1
    for bad in (0, 101, -3):
        assert lib.jpegamd_quality_tables(bad, a.ctypes.data, b.ctypes.data) != 0
        assert b"between 1 and 100" in ctypes.cast(lib.jpegamd_last_error(), ctypes.c_char_p).value


def test_harness_argument_and_input_errors(tmp_path):
    """main.cpp:28-47 (quality parsing) and readImage's failure, before any device call."""
    r = _run(["-q", "0"], "", tmp_path)
    assert r.returncode == 1 and "Invalid quality argument: Quality must be between 1 and 100" in r.stderr
    r = _run(["-q", "abc"], "", tmp_path)
    assert r.returncode == 1 and "Invalid quality argument" in r.stderr
    r = _run(["-q", "75"], "no_such_file.ppm\n", tmp_path)
    assert r.returncode == 1 and r.stdout.startswith("Using quality: 75%\nEnter the image name: ")
    assert "Could not open or find the image" in r.stderr
    _write_ppm(tmp_path / "odd.ppm", np.zeros((20, 32, 3), np.uint8))
    r = _run([], "odd.ppm\n", tmp_path)
    assert r.returncode == 1 and "multiples of 16" in r.stderr
    (tmp_path / "bad.ppm").write_bytes(b"P5\n16 16\n255\n" + bytes(256))
    r = _run([], "bad.ppm\n", tmp_path)
    assert r.returncode == 1 and "P6" in r.stderr
    (tmp_path / "short.ppm").write_bytes(b"P6\n16 16\n255\n" + bytes(100))
    r = _run([], "short.ppm\n", tmp_path)
    assert r.returncode == 1 and "truncated" in r.stderr


@pytest.mark.gpu
def test_harness_lena_quality_50(gpu, orc, lena_bgr, tmp_path):
    """config 1 through the command line: same .bin as the reference's committed file,
    the reference's report lines, PSNR 35.2317 dB within 0.01."""
    _write_ppm(tmp_path / "lena.ppm", lena_bgr)
    os.mkdir(tmp_path / "output")
    r = _run(["-q", "50"], "lena.ppm\n", tmp_path)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    for line in ("Using quality: 50%", "Image Dimensions: 1024 x 1024", "Compressed with GPU",
                 "Compressed file (GPU) saved in: output/compressed_image_gpu_50.bin",
                 "Original size (GPU): 3145728 bytes (3072 KB)", "Decompressed with GPU", "Image Metrics (GPU):"):
        assert line in out, line
    data = (tmp_path / "output" / "compressed_image_gpu_50.bin").read_bytes()
    assert data == open(os.path.join(GOLD, "compressed_image_cpu_50.bin"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == "7870d3fec500635c7ef9dc5a98f985f31dbb2c91b0beb8d0c6ba9f9dbf82b2bf"
    assert f"Compressed size (GPU): {len(data)} bytes" in out
    assert abs(float(re.search(r"Compression ratio \(GPU\): ([\d.]+)", out).group(1)) - 3145728 / len(data)) < 1e-3
    for what in ("Encoding", "Decoding"):
        assert float(re.search(what + r" time \(GPU\): ([\d.e+-]+) ms", out).group(1)) > 0
    psnr = float(re.search(r"PSNR: ([\d.]+) dB", out).group(1))
    mse = float(re.search(r"MSE: ([\d.]+)", out).group(1))
    assert abs(psnr - 35.2317) < 0.01
    rec = _read_ppm(tmp_path / "output" / "decompress_image_gpu_50.ppm")
    assert abs(orc.mse(lena_bgr, rec) - mse) < 1e-3 and abs(orc.psnr(orc.mse(lena_bgr, rec)) - psnr) < 1e-3
    import jpeg_amd
    assert np.array_equal(rec, jpeg_amd.decompress_image(data, *jpeg_amd.quality_tables(50)))


@pytest.mark.gpu
def test_harness_reads_the_reference_png_and_writes_png(gpu, lena_bgr, tmp_path):
    """main.cpp reads img/lena_color_1024.png with cv::imread and writes the reconstruction
    with cv::imwrite(".png") (:86, :300): the same file in, the committed golden .bin out, and
    a PNG of the reconstruction that PIL reads back as the library's own pixels."""
    import shutil
    from PIL import Image
    import jpeg_amd
    shutil.copy(os.path.join(GOLD, "lena_color_1024.png"), tmp_path / "lena_color_1024.png")
    os.mkdir(tmp_path / "output")
    r = _run(["-q", "50"], "lena_color_1024.png\n", tmp_path)
    assert r.returncode == 0, r.stderr
    assert "Image Dimensions: 1024 x 1024" in r.stdout
    data = (tmp_path / "output" / "compressed_image_gpu_50.bin").read_bytes()
    assert hashlib.sha256(data).hexdigest() == "7870d3fec500635c7ef9dc5a98f985f31dbb2c91b0beb8d0c6ba9f9dbf82b2bf"
    rec = np.array(Image.open(tmp_path / "output" / "decompress_image_gpu_50.png").convert("RGB"))[:, :, ::-1]
    assert np.array_equal(rec, jpeg_amd.decompress_image(data, *jpeg_amd.quality_tables(50)))
    assert abs(float(re.search(r"PSNR: ([\d.]+) dB", r.stdout).group(1)) - 35.2317) < 0.01
    # the same picture as an uncompressed TIFF (the format of the reference's other samples)
    Image.fromarray(np.ascontiguousarray(lena_bgr[:, :, ::-1])).save(tmp_path / "lena.tif", compression=None)
    os.mkdir(tmp_path / "tif")
    r = _run(["-q", "50", "-o", "tif"], "lena.tif\n", tmp_path)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "tif" / "compressed_image_gpu_50.bin").read_bytes() == data
    assert (tmp_path / "tif" / "decompress_image_gpu_50.png").exists()


@pytest.mark.gpu
def test_harness_other_quality_and_output_dir(gpu, tmp_path):
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    img = synth_image(48, 80, 7, "photo").numpy()
    _write_ppm(tmp_path / "a.ppm", img)
    os.mkdir(tmp_path / "res")
    r = _run(["-q", "90", "-o", "res"], "a.ppm\n", tmp_path)
    assert r.returncode == 0, r.stderr
    data = (tmp_path / "res" / "compressed_image_gpu_90.bin").read_bytes()
    assert data == jpeg_amd.compress_image(img, *jpeg_amd.quality_tables(90))
    r = _run(["-o", "missing_dir"], "a.ppm\n", tmp_path)
    assert r.returncode == 1 and "Cannot open file for writing" in r.stderr


# ---- jpegamd_analysis: the GPU leg of analysis.cpp (13 numbers, fixed table pairs) -------------
ANALYSIS = os.path.join(ROOT, "jpeg_amd", "jpegamd_analysis")


def test_analysis_tables_are_the_literal_pairs_of_analysis_cpp(orc):
    """analysis.cpp:43-90: no scaling; quality >= 90 selects its own pair (luma [5][5] == 12),
    anything else the base pair.  C ABI == Python table == oracle restatement."""
    import jpeg_amd
    from jpeg_amd import _lib
    lib = _lib.load()
    for q in (1, 10, 49, 50, 89, 90, 95, 100):
        a, b = np.zeros(64, np.int32), np.zeros(64, np.int32)
        assert lib.jpegamd_analysis_tables(q, a.ctypes.data, b.ctypes.data) == 0
        py, pc = jpeg_amd.analysis_tables(q)
        oy, oc = orc.analysis_tables(q)
        assert np.array_equal(a, py) and np.array_equal(b, pc) and np.array_equal(a, oy) and np.array_equal(b, oc)
        if q >= 90:
            assert a[5 * 8 + 5] == 12 and b[0] == 11
        else:
            assert np.array_equal(a, jpeg_amd.quality_tables(50)[0])
    a = np.zeros(64, np.int32)
    assert lib.jpegamd_analysis_tables(0, a.ctypes.data, a.ctypes.data) != 0


def test_analysis_usage_errors(tmp_path):
    r = subprocess.run([ANALYSIS], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert r.returncode == 1 and "Usage:" in r.stderr and "<image_path> [quality]" in r.stderr   # analysis.cpp:35
    r = subprocess.run([ANALYSIS, "x.png", "101"], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert r.returncode == 1 and "Invalid quality argument: Quality must be between 1 and 100" in r.stderr
    r = subprocess.run([ANALYSIS, "no_such.png"], capture_output=True, text=True, cwd=tmp_path, timeout=60)
    assert r.returncode == 1 and "Could not open or find the image" in r.stderr


@pytest.mark.gpu
def test_analysis_prints_thirteen_numbers_in_the_reference_positions(gpu, lena_bgr, tmp_path):
    """parsed exactly as metric_image_file.py:41-65 does (non-empty lines, float of the text after
    the last ':').  At quality 50 analysis.cpp's tables are main.cpp's, so the file is the golden."""
    import shutil
    shutil.copy(os.path.join(GOLD, "lena_color_1024.png"), tmp_path / "lena.png")
    os.mkdir(tmp_path / "output")
    r = subprocess.run([ANALYSIS, "lena.png", "50", "--cpu-encode-ms", "2993", "--cpu-decode-ms", "3236"],
                       capture_output=True, text=True, cwd=tmp_path, timeout=110)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 13
    v = [float(ln.split(":")[-1].strip()) for ln in lines]
    data = (tmp_path / "output" / "compressed_image_gpu.bin").read_bytes()
    assert hashlib.sha256(data).hexdigest() == "7870d3fec500635c7ef9dc5a98f985f31dbb2c91b0beb8d0c6ba9f9dbf82b2bf"
    assert v[0] > 1 and v[1] > 1                                   # gains against the given CPU times
    assert abs(v[5] - 3145728 / len(data)) < 1e-2                   # ratio GPU (csv: 23.96 for the 1024^2 Lena)
    assert abs(v[9] - 19.4944) < 0.05 and abs(v[10] - 35.2317) < 0.01   # MSE / PSNR GPU vs BASELINE.md's CPU values
    for i in (2, 3, 4, 6, 7, 8, 11, 12):                            # CPU / OMP slots: no such back-end here
        assert v[i] != v[i]
    assert (tmp_path / "output" / "decompress_image_gpu.png").exists()
    # quality >= 90 switches to the other fixed pair
    r = subprocess.run([ANALYSIS, "lena.png", "95"], capture_output=True, text=True, cwd=tmp_path, timeout=110)
    assert r.returncode == 0, r.stderr
    hi = [float(ln.split(":")[-1].strip()) for ln in r.stdout.splitlines() if ln.strip()]
    assert hi[5] < v[5] and hi[10] > v[10] and hi[0] != hi[0]
    import jpeg_amd
    assert (tmp_path / "output" / "compressed_image_gpu.bin").read_bytes() == \
        jpeg_amd.compress_image(lena_bgr, *jpeg_amd.analysis_tables(95))
