"""The measurement configurations of SURVEY.md 8d at sizes the oracle finishes in
seconds: the quality sweep (config 5), the gray / odd-size case (config 3) and the
batch of independent images (config 4).  Same bars as tests/test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _check_coef, _check_plane_px, _check_px, _t

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("qf", list(range(1, 101)))
def test_quality_sweep(gpu, orc, qf):
    """config 5, every quality factor 1..100 ("parity at every QF"): every branch of the table scaling (main.cpp:51-80: qf < 50, qf >= 50,
    the clamps to 1 and to 255) through encode -> .bin -> decode against the oracle."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    qy, qc = jpeg_amd.quality_tables(qf)
    oy, oc = orc.quality_tables(qf)
    assert np.array_equal(qy, oy) and np.array_equal(qc, oc)
    bgr = synth_image(128, 256, 1000 + qf, "photo").numpy()
    want = orc.encode_image(bgr, qy, qc)
    got = [c.cpu().numpy() for c in jpeg_amd.encode_image(_t(bgr, gpu), qy, qc)]
    same = True
    for g, w, n in zip(got, want, "Y Cb Cr".split()):
        same &= _check_coef(g, w, f"QF={qf} {n}") == 0
    data = jpeg_amd.compress_image(bgr, qy, qc)
    if same:  # then the file is the reference's, byte for byte
        assert data == orc.write_bin(*(orc.rle_stream(c) for c in want), 128, 256)
    rec = jpeg_amd.decompress_image(data, qy, qc)
    # the decoded planes against the oracle's: +-1, differences at exact x.5 values counted apart
    # (low quality factors make flat blocks whose pixels sit exactly on x.5: _check_plane_px) ...
    planes = []
    for g, (r, c), q, n in zip(got, ((128, 256), (64, 128), (64, 128)), (qy, qc, qc), "Y Cb Cr".split()):
        pl = jpeg_amd.decode_plane(_t(g, gpu), r, c, q).cpu().numpy()
        _check_plane_px(pl, orc.decode_plane(g, r, c, q), g, q, f"QF={qf} {n} plane")
        planes.append(pl)
    # ... and the image out of the file is, bit for bit, the reference's colour stage on those planes
    up = lambda c: np.repeat(np.repeat(c, 2, axis=0), 2, axis=1)
    ycc = np.ascontiguousarray(np.stack([planes[0], up(planes[1]), up(planes[2])], -1))
    assert np.array_equal(rec, orc.ycbcr2rgb(ycc)), f"QF={qf}: decoded image is not YCbCr2RGB of the decoded planes"
    # north_star: "MSE/PSNR identical to 0.01 dB" -- the GPU's image against the reference path's image
    # (the oracle decoding the oracle's own coefficients), at every quality factor
    ref_img = orc.decode_image(*want, 128, 256, qy, qc)
    p_gpu, p_ref = orc.psnr(orc.mse(bgr, rec)), orc.psnr(orc.mse(bgr, ref_img))
    assert abs(p_gpu - p_ref) < 0.01, f"QF={qf}: PSNR {p_gpu:.4f} dB against the reference path's {p_ref:.4f} dB"


def test_gray_image_has_flat_chroma(gpu, orc):
    """config 3: B = G = R.  Y equals the gray value, Cb = Cr = 128, so every chroma
    block is all zero after the level shift and its run-length stream is (0, 64)."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    rows, cols = 112, 336
    qy, qc = jpeg_amd.quality_tables(90)
    bgr = synth_image(rows, cols, 3, "gray").numpy()
    assert np.array_equal(bgr[..., 0], bgr[..., 1]) and np.array_equal(bgr[..., 1], bgr[..., 2])
    cy, ccb, ccr = jpeg_amd.encode_image(_t(bgr, gpu), qy, qc)
    assert int(ccb.abs().sum()) == 0 and int(ccr.abs().sum()) == 0
    pairs, offs = jpeg_amd.rle_encode_plane(ccb)
    assert pairs.shape[0] == ccb.shape[0] and bool((pairs[:, 0] == 0).all()) and bool((pairs[:, 1] == 64).all())
    for g, w, n in zip((cy, ccb, ccr), orc.encode_image(bgr, qy, qc), "Y Cb Cr".split()):
        _check_coef(g.cpu().numpy(), w, n)
    rec = jpeg_amd.decode_image(cy, ccb, ccr, rows, cols, qy, qc).cpu().numpy()
    assert np.array_equal(rec[..., 0], rec[..., 1]) and np.array_equal(rec[..., 1], rec[..., 2])


def test_odd_plane_padding_branch(gpu, orc):
    """config 3, second half: the zero-fill / drop branch of extractBlocks and the block
    scatter (encode_decode.cpp:248-251, 702-703) is only reachable at the plane level."""
    import jpeg_amd
    rng = np.random.default_rng(33)
    for rows, cols in ((3580 // 20, 5370 // 30), (1, 1), (7, 9), (8, 8), (9, 7), (17, 515)):
        q = jpeg_amd.quality_tables(90)[0]
        plane = rng.integers(0, 256, size=(rows, cols), dtype=np.uint8)
        want, _ = orc.encode_plane(plane, q, want_symbols=False)
        got = jpeg_amd.encode_plane(_t(plane, gpu), q).cpu().numpy()
        _check_coef(got, want, f"{rows}x{cols} plane")
        rec = jpeg_amd.decode_plane(_t(want, gpu), rows, cols, q).cpu().numpy()
        _check_px(rec, orc.decode_plane(want, rows, cols, q), f"{rows}x{cols} plane decode")


def test_batch_of_independent_images(gpu, orc):
    """config 4: images are independent (no state survives a call): a batch encoded
    back to back on one stream, and interleaved on two streams, equals one at a time."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    qy, qc = jpeg_amd.quality_tables(50)
    rows, cols = 144, 256  # 3840x2160 scaled down by 15
    imgs = [synth_image(rows, cols, seed, "photo", device=gpu) for seed in range(6)]
    alone = []
    for im in imgs:
        alone.append([c.clone() for c in jpeg_amd.encode_image(im, qy, qc)])
        torch.cuda.synchronize()
    batch = [jpeg_amd.encode_image(im, qy, qc) for im in imgs]
    torch.cuda.synchronize()
    for a, b in zip(alone, batch):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for i, im in enumerate(imgs):
        with torch.cuda.stream(s1 if i % 2 else s2):
            outs.append(jpeg_amd.encode_image(im, qy, qc))
    torch.cuda.synchronize()
    for a, b in zip(alone, outs):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    want = orc.encode_image(imgs[0].cpu().numpy(), qy, qc)
    for g, w, n in zip(alone[0], want, "Y Cb Cr".split()):
        _check_coef(g.cpu().numpy(), w, n)


def _strip_parity(orc, bgr, coefs, rec, qy, qc, mcu_rows, what):
    """every 16x16 MCU is independent: sampled MCU-row strips of a full-size image against
    the oracle (coefficients and decoded pixels)"""
    rows, cols = bgr.shape[:2]
    bw, bwc = cols // 8, cols // 16
    cy, cb, cr = coefs
    for my in mcu_rows:
        strip = bgr[my * 16:(my + 1) * 16].cpu().numpy()
        wy, wcb, wcr = orc.encode_image(strip, qy, qc)
        _check_coef(cy[my * 2 * bw:(my * 2 + 2) * bw].cpu().numpy(), wy, f"{what} strip {my} Y")
        _check_coef(cb[my * bwc:(my + 1) * bwc].cpu().numpy(), wcb, f"{what} strip {my} Cb")
        _check_coef(cr[my * bwc:(my + 1) * bwc].cpu().numpy(), wcr, f"{what} strip {my} Cr")
        _check_px(rec[my * 16:(my + 1) * 16].cpu().numpy(), orc.decode_image(wy, wcb, wcr, 16, cols, qy, qc),
                  f"{what} strip {my} pixels", lsb=2)


def test_config3_full_size_gray(gpu, orc):
    """config 3 at its real size: 3584 x 5376, B = G = R, QF = 90 (the reference's gray file
    is absent, .MISSING_LARGE_BLOBS; imread would hand over three equal channels): chroma
    planes are all zero, Y strips match the oracle, and the file round-trips."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    rows, cols = 3584, 5376
    bgr = synth_image(rows, cols, 33, "gray", device=gpu)
    assert bool((bgr[..., 0] == bgr[..., 1]).all()) and bool((bgr[..., 1] == bgr[..., 2]).all())
    qy, qc = jpeg_amd.quality_tables(90)
    coefs = jpeg_amd.encode_image(bgr, qy, qc)
    assert not bool(coefs[1].any()) and not bool(coefs[2].any())  # Cb = Cr = 128 everywhere
    rec = jpeg_amd.decode_image(*coefs, rows, cols, qy, qc)
    assert bool((rec[..., 0] == rec[..., 1]).all()) and bool((rec[..., 1] == rec[..., 2]).all())
    _strip_parity(orc, bgr, coefs, rec, qy, qc, (0, 100, rows // 16 - 1), "gray")
    data = jpeg_amd.compress_image(bgr.cpu().numpy(), qy, qc)
    back = jpeg_amd.decompress_image(data, qy, qc)
    assert np.array_equal(back, rec.cpu().numpy())


def test_batch_in_one_launch_is_the_images_one_by_one(gpu):
    """config 4 for small images: a contiguous batch of equally sized images is one image of
    n * rows rows (MCU rows are independent), so it goes through ONE encode and ONE decode
    launch; coefficients and pixels equal the per-image calls bit for bit."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    rows, cols, n = 48, 528, 5
    qy, qc = jpeg_amd.quality_tables(75)
    batch = torch.stack([synth_image(rows, cols, 70 + i, "noise" if i == 2 else "photo", device=gpu) for i in range(n)])
    cy, cb, cr = jpeg_amd.encode_images(batch, qy, qc)
    assert cy.shape == (n, rows * cols // 64, 64) and cb.shape == (n, rows * cols // 256, 64)
    rec = jpeg_amd.decode_images(cy, cb, cr, rows, cols, qy, qc)
    for i in range(n):
        one = jpeg_amd.encode_image(batch[i], qy, qc)
        assert torch.equal(one[0], cy[i]) and torch.equal(one[1], cb[i]) and torch.equal(one[2], cr[i])
        assert torch.equal(jpeg_amd.decode_image(*one, rows, cols, qy, qc), rec[i])


def test_config4_full_size_batch(gpu, orc):
    """config 4 at its real image size (3840 x 2160: 7.5 wave tiles per row, so every MCU row
    ends in a half-filled tile): images of a batch processed back to back on two streams equal
    the same images processed alone, and sampled strips match the oracle."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    rows, cols = 2160, 3840
    qy, qc = jpeg_amd.quality_tables(50)
    imgs = [synth_image(rows, cols, seed, "photo", device=gpu) for seed in range(4)]
    alone = [jpeg_amd.encode_image(im, qy, qc) for im in imgs]
    recs = [jpeg_amd.decode_image(*c, rows, cols, qy, qc) for c in alone]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=gpu) for _ in range(2)]
    outs, recs2 = [], []
    for i, im in enumerate(imgs):
        with torch.cuda.stream(streams[i & 1]):
            c = jpeg_amd.encode_image(im, qy, qc)
            outs.append(c)
            recs2.append(jpeg_amd.decode_image(*c, rows, cols, qy, qc))
    torch.cuda.synchronize()
    for a, b, r, r2 in zip(alone, outs, recs, recs2):
        assert all(torch.equal(x, y) for x, y in zip(a, b)) and torch.equal(r, r2)
    _strip_parity(orc, imgs[3], alone[3], recs[3], qy, qc, (0, 67, rows // 16 - 1), "batch image 3")


@pytest.mark.parametrize("qf", [1, 50, 100])
def test_config5_full_size_quality(gpu, orc, qf):
    """config 5 at its real size (4096^2) at both ends and the middle of the quality range"""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    rows = cols = 4096
    bgr = synth_image(rows, cols, 4096 + qf, "photo", device=gpu)
    qy, qc = jpeg_amd.quality_tables(qf)
    coefs = jpeg_amd.encode_image(bgr, qy, qc)
    rec = jpeg_amd.decode_image(*coefs, rows, cols, qy, qc)
    _strip_parity(orc, bgr, coefs, rec, qy, qc, (0, 129, 255), f"QF={qf}")
    again = jpeg_amd.encode_image(bgr, qy, qc)
    assert all(torch.equal(x, y) for x, y in zip(coefs, again))  # deterministic


@pytest.mark.parametrize("qf", [10, 50, 90])
def test_config5_lena_upscaled(gpu, orc, qf):
    """config 5 on ITS input: the Lena photograph x4 (integer bilinear, jpeg_amd.synth.lena_upscaled:
    the committed 1024^2 golden input, no library resize filter) = 4096^2.  Strip parity against the
    oracle, the file round trip, and PSNR of the GPU's image against the reference path's on the strips."""
    import jpeg_amd
    from jpeg_amd.synth import lena_upscaled
    bgr = lena_upscaled(4, device=gpu)
    rows, cols = int(bgr.shape[0]), int(bgr.shape[1])
    assert (rows, cols) == (4096, 4096)
    qy, qc = jpeg_amd.quality_tables(qf)
    coefs = jpeg_amd.encode_image(bgr, qy, qc)
    rec = jpeg_amd.decode_image(*coefs, rows, cols, qy, qc)
    mcu_rows = (0, 77, 128, 255)
    _strip_parity(orc, bgr, coefs, rec, qy, qc, mcu_rows, f"Lena x4 QF={qf}")
    # PSNR over the sampled strips: the GPU's pixels against the reference path's (oracle end to end)
    src = np.concatenate([bgr[my * 16:(my + 1) * 16].cpu().numpy() for my in mcu_rows])
    got = np.concatenate([rec[my * 16:(my + 1) * 16].cpu().numpy() for my in mcu_rows])
    ref = np.concatenate([orc.decode_image(*orc.encode_image(bgr[my * 16:(my + 1) * 16].cpu().numpy(), qy, qc), 16, cols, qy, qc)
                          for my in mcu_rows])
    assert abs(orc.psnr(orc.mse(src, got)) - orc.psnr(orc.mse(src, ref))) < 0.01
    if qf == 50:
        data = jpeg_amd.compress_image(bgr.cpu().numpy(), qy, qc)
        assert np.array_equal(jpeg_amd.decompress_image(data, qy, qc), rec.cpu().numpy())


def test_device_entry_points_can_be_captured_in_a_hip_graph(gpu):
    """the _dev entry points allocate nothing and never synchronise, so a caller may record
    them into a HIP graph and replay it (serving loops do); results equal the eager ones"""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    rows, cols = 64, 1040
    qy, qc = jpeg_amd.quality_tables(60)
    img = synth_image(rows, cols, 5, "photo", device=gpu)
    eager = jpeg_amd.encode_image(img, qy, qc)
    eager_rec = jpeg_amd.decode_image(*eager, rows, cols, qy, qc)
    coefs = tuple(torch.zeros_like(c) for c in eager)
    rec = torch.zeros_like(eager_rec)
    side = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(side):  # warm-up on the capture stream, as torch asks
        jpeg_amd.encode_image(img, qy, qc, out=coefs, stream=side.cuda_stream)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        jpeg_amd.encode_image(img, qy, qc, out=coefs, stream=side.cuda_stream)
        jpeg_amd.decode_image(*coefs, rows, cols, qy, qc, out=rec, stream=side.cuda_stream)
    for c in coefs:
        c.zero_()
    rec.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(coefs, eager)) and torch.equal(rec, eager_rec)
