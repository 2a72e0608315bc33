"""Whole-image host entry points (jpegamd_compress_image / jpegamd_decompress_image):
the reference's main.cpp "GPU" branch end to end -- image -> .bin -> image -- against
the committed golden files of config 1 (lena_color_1024.png, QF=50)."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_lena_to_bin_and_back(gpu, orc, lena_bgr):
    import jpeg_amd
    from PIL import Image
    qy, qc = jpeg_amd.quality_tables(50)
    data = jpeg_amd.compress_image(lena_bgr, qy, qc)
    gold = open(os.path.join(GOLD, "compressed_image_cpu_50.bin"), "rb").read()
    assert data == gold, "the .bin written through the GPU path differs from the reference's"
    assert hashlib.sha256(data).hexdigest() == "7870d3fec500635c7ef9dc5a98f985f31dbb2c91b0beb8d0c6ba9f9dbf82b2bf"
    rec = jpeg_amd.decompress_image(gold, qy, qc)
    gold_px = np.array(Image.open(os.path.join(GOLD, "decompress_image_cpu_50.png")).convert("RGB"))[:, :, ::-1]
    assert rec.shape == gold_px.shape
    d = np.abs(rec.astype(int) - gold_px.astype(int))
    # the IDCT is float work: +-1 on a plane sample, which the chroma gains stretch to 2
    assert d.max() <= 2 and (d != 0).sum() <= 480e-6 * d.size
    assert abs(orc.psnr(orc.mse(lena_bgr, rec)) - 35.2317) < 0.01


def test_round_trip_other_sizes_and_tables(gpu, orc):
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    for rows, cols, qf, kind in ((16, 16, 50, "noise"), (48, 1040, 10, "photo"), (528, 64, 95, "photo"),
                                 (64, 512, 90, "gray")):
        qy, qc = jpeg_amd.quality_tables(qf)
        img = synth_image(rows, cols, rows + cols, kind).numpy()
        data = jpeg_amd.compress_image(img, qy, qc)
        # same bytes as: fused encode on the device -> oracle RLE + entropy coder
        import torch
        coefs = [c.cpu().numpy() for c in jpeg_amd.encode_image(torch.from_numpy(img).to(gpu), qy, qc)]
        assert data == orc.write_bin(*(orc.rle_stream(c) for c in coefs), rows, cols)
        rec = jpeg_amd.decompress_image(data, qy, qc)
        want = jpeg_amd.decode_image(*(torch.from_numpy(c).to(gpu) for c in coefs), rows, cols, qy, qc).cpu().numpy()
        assert np.array_equal(rec, want)


def test_pipeline_errors(gpu):
    import jpeg_amd
    qy, qc = jpeg_amd.quality_tables(50)
    with pytest.raises(jpeg_amd.JpegAmdError, match="multiples of 16"):
        jpeg_amd.compress_image(np.zeros((20, 16, 3), np.uint8), qy, qc)
    with pytest.raises(jpeg_amd.JpegAmdError):
        jpeg_amd.decompress_image(b"\x00" * 20, qy, qc)


def test_crafted_bin_with_oversized_runs_returns_quickly(gpu, orc):
    """decompress_image accepts arbitrary bytes.  A .bin whose pair streams carry
    runs of 2^31-1 (built with the oracle's container writer) must decode in bounded time with
    the overflow dropped: plane = what the reference's expand-then-cut would keep."""
    import time
    import torch
    import jpeg_amd
    rows, cols = 64, 96
    big = 2**31 - 1
    nby, nbc = (rows // 8) * (cols // 8), (rows // 16) * (cols // 16)
    sy = np.array([7, 3, 0, 61, -2, big, 9, big, 5, 64], np.int32)     # block 0, then -2 to the end of the plane
    scb = np.array([1, big] * 200, np.int32)                             # every pair oversized
    scr = np.array([0, 64] * nbc, np.int32)
    data = orc.write_bin(sy, scb, scr, rows, cols)
    qy, qc = jpeg_amd.quality_tables(50)
    t0 = time.perf_counter()
    rec = jpeg_amd.decompress_image(data, qy, qc)
    assert time.perf_counter() - t0 < 10.0
    cy = np.full((nby, 64), -2, np.int16)
    cy[0, :3], cy[0, 3:] = 7, 0
    ccb = np.full((nbc, 64), 1, np.int16)
    ccr = np.zeros((nbc, 64), np.int16)
    want = jpeg_amd.decode_image(*(torch.from_numpy(c).to(gpu) for c in (cy, ccb, ccr)), rows, cols, qy, qc)
    assert np.array_equal(rec, want.cpu().numpy())


def test_host_pointer_entry_points_with_row_pitch(gpu):
    """jpegamd_encode_image / jpegamd_decode_image on pageable host memory (the level-1 boundary
    of INTEGRATION.md): large enough for the staged copier, and with a row pitch wider than the
    image, like a cv::Mat ROI.  Must equal the device-pointer path bit for bit."""
    import torch
    import jpeg_amd
    from jpeg_amd import _lib
    from jpeg_amd.synth import synth_image
    lib = _lib.load()
    rows, cols, pitch = 1040, 2064, 2064 * 3 + 52
    img = synth_image(rows, cols, 77, "photo").numpy()
    padded = np.full((rows, pitch), 0xAB, np.uint8)
    padded[:, :cols * 3] = img.reshape(rows, cols * 3)
    qy, qc = jpeg_amd.quality_tables(75)
    qy32, qc32 = np.ascontiguousarray(qy, np.int32), np.ascontiguousarray(qc, np.int32)
    ny, nc = rows * cols, rows * cols // 4
    cy, cb, cr = np.empty(ny, np.int16), np.empty(nc, np.int16), np.empty(nc, np.int16)
    _lib.check(lib.jpegamd_encode_image(padded.ctypes.data, rows, cols, pitch, qy32.ctypes.data, qc32.ctypes.data,
                                        cy.ctypes.data, cb.ctypes.data, cr.ctypes.data))
    want = jpeg_amd.encode_image(torch.from_numpy(img).to(gpu), qy, qc)
    for g, w in zip((cy, cb, cr), want):
        assert np.array_equal(g.reshape(-1, 64), w.cpu().numpy())
    out = np.full((rows, pitch), 0xCD, np.uint8)
    _lib.check(lib.jpegamd_decode_image(cy.ctypes.data, cb.ctypes.data, cr.ctypes.data, rows, cols,
                                        qy32.ctypes.data, qc32.ctypes.data, out.ctypes.data, pitch))
    rec = jpeg_amd.decode_image(*want, rows, cols, qy, qc).cpu().numpy()
    assert np.array_equal(out[:, :cols * 3].reshape(rows, cols, 3), rec)
    assert (out[:, cols * 3:] == 0xCD).all()  # the padding of every row is left alone


def test_context_keeps_buffers_and_gives_the_same_bytes(gpu):
    """jpegamd_context_*: the host-pointer calls with device buffers kept between calls
    (SURVEY 8b).  Images of growing and shrinking sizes through one context give the bytes
    and pixels of the plain calls; a second context on the same device is independent."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    qy, qc = jpeg_amd.quality_tables(60)
    with jpeg_amd.Context(0) as ctx, jpeg_amd.Context(0) as other:
        for i, (rows, cols, kind) in enumerate([(64, 528, "photo"), (256, 1024, "noise"), (32, 48, "photo"),
                                                (256, 1024, "photo"), (16, 16, "gray")]):
            img = synth_image(rows, cols, 40 + i, kind).numpy()
            want = jpeg_amd.compress_image(img, qy, qc)
            got = ctx.compress_image(img, qy, qc)
            assert got == want, (rows, cols)
            rec = ctx.decompress_image(got, qy, qc)
            assert np.array_equal(rec, jpeg_amd.decompress_image(want, qy, qc))
            if i == 1:
                assert other.compress_image(img, qy, qc) == want
    with pytest.raises(jpeg_amd.JpegAmdError):
        jpeg_amd.Context(99)
