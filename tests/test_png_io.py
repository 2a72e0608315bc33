"""The command-line harness's PNG reader and writer (jpeg_amd/csrc/host/png_io.h, zlib only)
against PIL: what cv::imread(IMREAD_COLOR) makes of the kinds of PNG the reference ships, and
files PIL can read back.  Host code only."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "png_io_driver")


def _decode(path, tmp_path):
    raw = tmp_path / "out.raw"
    r = subprocess.run([DRIVER, "decode", str(path), str(raw)], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr.strip())
    data = raw.read_bytes()
    rows, cols = np.frombuffer(data[:8], np.int32)
    return np.frombuffer(data[8:], np.uint8).reshape(rows, cols, 3)


def test_lena_golden_png(tmp_path, lena_bgr):
    got = _decode(os.path.join(ROOT, "tests", "golden", "lena_color_1024.png"), tmp_path)
    assert np.array_equal(got, lena_bgr)


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1", "I;16"])
def test_colour_types_and_filters(tmp_path, mode):
    rng = np.random.default_rng(3)
    h, w = 37, 53
    smooth = (np.add.outer(np.arange(h) * 5, np.arange(w) * 3)[..., None] + rng.integers(0, 9, (h, w, 3))) % 256
    rgb = smooth.astype(np.uint8)
    if mode == "I;16":
        g16 = (rgb[..., 0].astype(np.uint16) << 8) | rgb[..., 1]
        im = Image.fromarray(g16)
        want = np.repeat((g16 >> 8).astype(np.uint8)[..., None], 3, -1)
    else:
        im = Image.fromarray(rgb, "RGB").convert(mode)
        want = np.array(im.convert("RGB"))
    path = tmp_path / f"{mode.replace(';', '')}.png"
    im.save(path, optimize=True)  # lets the encoder choose among the five scanline filters
    got = _decode(path, tmp_path)
    assert np.array_equal(got[:, :, ::-1], want)


@pytest.mark.parametrize("mode,kw", [("RGB", {}), ("L", {}), ("RGBA", {}), ("RGB", {"big": True})])
def test_uncompressed_tiff(tmp_path, mode, kw):
    """the reference's img/*.tif are uncompressed 8-bit grey or RGB files"""
    rng = np.random.default_rng(8)
    rgb = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    im = Image.fromarray(rgb, "RGB").convert(mode)
    path = tmp_path / "a.tif"
    im.save(path, compression=None)
    if kw.get("big"):  # the same file in the other byte order, written by hand
        w, h = im.size
        data = np.array(im).tobytes()
        tags = [(256, 3, 1, w), (257, 3, 1, h), (258, 3, 1, 8), (259, 3, 1, 1), (262, 3, 1, 2), (273, 4, 1, 8),
                (277, 3, 1, 3), (278, 3, 1, h), (279, 4, 1, len(data))]
        ifd_at = 8 + len(data) + (len(data) & 1)
        ifd = len(tags).to_bytes(2, "big")
        for tag, typ, cnt, val in tags:
            ifd += tag.to_bytes(2, "big") + typ.to_bytes(2, "big") + cnt.to_bytes(4, "big")
            ifd += val.to_bytes(2, "big") + b"\0\0" if typ == 3 else val.to_bytes(4, "big")
        path.write_bytes(b"MM\0*" + ifd_at.to_bytes(4, "big") + data + b"\0" * (len(data) & 1) + ifd + b"\0\0\0\0")
        # (bits per sample 8 given once for three samples: readers take it for all of them)
    got = _decode(path, tmp_path)
    assert np.array_equal(got[:, :, ::-1], np.array(Image.open(path).convert("RGB")))


def test_compressed_tiff_is_refused(tmp_path):
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(tmp_path / "z.tif", compression="tiff_lzw")
    with pytest.raises(RuntimeError, match="compressed TIFF"):
        _decode(tmp_path / "z.tif", tmp_path)


def test_writer_is_read_back_by_pil(tmp_path):
    rng = np.random.default_rng(4)
    bgr = rng.integers(0, 256, (48, 80, 3), dtype=np.uint8)
    raw = tmp_path / "in.raw"
    raw.write_bytes(np.array(bgr.shape[:2], np.int32).tobytes() + bgr.tobytes())
    out = tmp_path / "out.png"
    subprocess.run([DRIVER, "encode", str(raw), str(out)], check=True)
    assert np.array_equal(np.array(Image.open(out).convert("RGB"))[:, :, ::-1], bgr)
    assert np.array_equal(_decode(out, tmp_path), bgr)


def test_bad_files_fail_loudly(tmp_path):
    good = open(os.path.join(ROOT, "tests", "golden", "lena_color_1024.png"), "rb").read()
    (tmp_path / "cut.png").write_bytes(good[:5000])
    with pytest.raises(RuntimeError):
        _decode(tmp_path / "cut.png", tmp_path)
    flipped = bytearray(good)
    flipped[200] ^= 1
    (tmp_path / "flip.png").write_bytes(bytes(flipped))
    with pytest.raises(RuntimeError, match="checksum"):
        _decode(tmp_path / "flip.png", tmp_path)
