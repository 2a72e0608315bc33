"""Device run-length stage (jpeg_amd/csrc/rle_kernels.hip) against the oracle's
run_length_encode / decodeRLE restatement (encode_decode.cpp:350-373, 484-518).
Integer work: bit-exact everywhere."""
import hashlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _planes(rng):
    """zigzag coefficient planes covering the shapes of run structure"""
    def sparse(nb, keep):
        c = np.zeros((nb, 64), np.int16)
        m = rng.random((nb, 64)) < keep * np.linspace(1.0, 0.02, 64)[None, :]
        c[m] = rng.integers(-30, 31, size=int(m.sum()), dtype=np.int16)
        c[:, 0] = rng.integers(-1024, 1025, size=nb, dtype=np.int16)
        return c
    alt = np.tile(np.array([5, -5], np.int16), (70, 32))             # 64 pairs per block
    const = np.full((130, 64), -7, np.int16)                            # 1 pair per block
    yield "one block", sparse(1, 0.5)
    yield "63 blocks", sparse(63, 0.5)
    yield "257 blocks", sparse(257, 0.3)
    yield "1000 sparse", sparse(1000, 0.15)
    yield "4099 dense", rng.integers(-900, 901, size=(4099, 64), dtype=np.int16)
    yield "all zero", np.zeros((300, 64), np.int16)
    yield "alternating", alt
    yield "constant", const
    yield "int16 extremes", np.tile(np.array([-32768, 32767, 32767, 0], np.int16), (65, 16))


def test_rle_encode_matches_oracle(gpu, orc):
    import jpeg_amd
    rng = np.random.default_rng(2024)
    for name, coef in _planes(rng):
        pairs, offs = jpeg_amd.rle_encode_plane(torch.from_numpy(coef).to(gpu))
        want = orc.rle_stream(coef)
        got = pairs.cpu().numpy().reshape(-1)
        assert np.array_equal(got, want), name
        offs = offs.cpu().numpy()
        counts = np.diff(offs)
        assert offs[0] == 0 and offs[-1] * 2 == want.size, name
        # per block: the pairs of block b are exactly run_length_encode(block b), runs sum to 64
        runs = got[1::2]
        assert np.array_equal(np.add.reduceat(runs, offs[:-1]), np.full(coef.shape[0], 64)), name
        assert counts.min() >= 1 and counts.max() <= 64, name


def test_rle_encode_three_planes_in_one_launch(gpu, orc):
    """jpegamd_rle_encode_image_dev: one launch set for three planes; the emit kernel's running sum of tile
    totals restarts at every plane's first tile and the plane's total is written by the workgroup of its last
    one; planes of very different sizes, bit-exact vs the oracle and the per-plane call."""
    import jpeg_amd
    rng = np.random.default_rng(5)
    def sparse(nb):
        c = np.zeros((nb, 64), np.int16)
        m = rng.random((nb, 64)) < 0.15
        c[m] = rng.integers(-30, 31, size=int(m.sum()), dtype=np.int16)
        return c
    for nbs in ((70000, 300, 1), (1, 1, 1), (257, 256, 255), (5, 20000, 513)):
        planes = [sparse(nb) for nb in nbs]
        dev = [torch.from_numpy(p).to(gpu) for p in planes]
        got = jpeg_amd.rle_encode_planes(dev)
        for (pairs, offs), p, d in zip(got, planes, dev):
            assert np.array_equal(pairs.cpu().numpy().reshape(-1), orc.rle_stream(p)), nbs
            one_pairs, one_offs = jpeg_amd.rle_encode_plane(d)
            assert torch.equal(one_pairs, pairs) and torch.equal(one_offs, offs)


def test_fused_encode_with_run_length_stage(gpu, orc):
    """jpegamd_encode_image_rle_dev: the encoder's per-block pair counts feed the run-length
    kernels.  Same coefficient planes as the plain encoder, same pairs and offsets as the
    separate run-length call, and the oracle's stream; ragged tile widths, dense and sparse."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    for kind, seed, rows, cols, qf in (("photo", 51, 256, 1040, 50), ("noise", 52, 64, 528, 100), ("photo", 53, 16, 16, 10),
                                       ("gray", 54, 48, 2064, 90), ("noise", 55, 1024, 1024, 50)):
        qy, qc = jpeg_amd.quality_tables(qf)
        img = synth_image(rows, cols, seed, kind, device=gpu)
        coefs, streams = jpeg_amd.encode_image_rle(img, qy, qc)
        plain = jpeg_amd.encode_image(img, qy, qc)
        for c, p in zip(coefs, plain):
            assert torch.equal(c, p), (kind, rows, cols)
        for (pairs, offs), (want_pairs, want_offs), c in zip(streams, jpeg_amd.rle_encode_planes(list(plain)), plain):
            assert torch.equal(pairs, want_pairs) and torch.equal(offs, want_offs), (kind, rows, cols)
            assert np.array_equal(pairs.cpu().numpy().reshape(-1), orc.rle_stream(c.cpu().numpy()))


def test_rle_decode_matches_oracle(gpu, orc):
    import jpeg_amd
    rng = np.random.default_rng(7)
    for name, coef in _planes(rng):
        nb = coef.shape[0]
        sym = orc.rle_stream(coef)
        out = jpeg_amd.rle_decode_plane(torch.from_numpy(sym.reshape(-1, 2)).to(gpu), nb).cpu().numpy()
        want, done = orc.decode_rle(sym, nb)
        assert done == nb and np.array_equal(want, coef)
        assert np.array_equal(out, coef), name


def test_rle_decode_loader_quirks(gpu, orc):
    """trailing pad symbols beyond the last block are dropped, a short
    stream leaves zero blocks, non-positive runs expand to nothing, a run may
    straddle the cut every 64 values."""
    import jpeg_amd
    nb = 5
    sym = np.array([3, 40, -2, 30, 9, 64, 0, 0, 4, -3, 1, 100, 7, 80, 8, 50], np.int32)  # 364 values > 320
    out = jpeg_amd.rle_decode_plane(torch.from_numpy(sym.reshape(-1, 2)).to(gpu), nb).cpu().numpy()
    want, done = orc.decode_rle(sym, nb)
    assert done == nb and np.array_equal(out, want)
    short = np.array([1, 10, 0, 20, 5, 40], np.int32)   # 70 values: one block and a bit
    out = jpeg_amd.rle_decode_plane(torch.from_numpy(short.reshape(-1, 2)).to(gpu), nb).cpu().numpy()
    want = np.zeros((nb, 64), np.int16)
    flat = want.reshape(-1)
    flat[:10], flat[30:70] = 1, 5
    assert np.array_equal(out, want)
    empty = jpeg_amd.rle_decode_plane(torch.empty((0, 2), dtype=torch.int32, device=gpu), 3)
    assert int(empty.abs().sum()) == 0


def _expand(sym, nb):
    """decodeRLE (encode_decode.cpp:484-518) + the drop of everything beyond the plane (:702),
    with Python integers: the expected result for run lengths no vector could hold"""
    flat = np.zeros(nb * 64, np.int16)
    pos = 0
    for v, r in np.asarray(sym, np.int64).reshape(-1, 2):
        if r > 0:
            if pos < flat.size:
                flat[pos:min(pos + int(r), flat.size)] = v
            pos += int(r)
    return flat.reshape(nb, 64)


def test_rle_decode_untrusted_run_lengths(gpu):
    """runs come from a file.  Huge and negative runs must neither wrap positions
    (values landing in the wrong blocks) nor keep the GPU busy: the overflow is dropped."""
    import time
    import jpeg_amd
    big = 2**31 - 1
    cases = [
        (3, [5, big]),
        (3, [1, 10, 7, big, 9, 5]),
        (4, [0, big, 3, 4]),
        (4, [2, 100, 4, -5, 6, big, 8, big, 9, 1]),
        (2, [1, 127, 2, 2**31 - 1, 3, 2**31 - 1]),          # uint32 sums would wrap back into the plane
        (5, [4, 65, 0, 70, -3, 200]),                          # long runs inside the plane (wave fill)
    ]
    for nb, sym in cases:
        sym = np.array(sym, np.int32)
        out = jpeg_amd.rle_decode_plane(torch.from_numpy(sym.reshape(-1, 2)).to(gpu), nb).cpu().numpy()
        assert np.array_equal(out, _expand(sym, nb)), (nb, sym)
    # a 1024^2 luma plane's worth of blocks and a million pairs with huge runs: seconds, not hours
    nb = 16384
    n = nb * 64
    sym = np.empty((n, 2), np.int32)
    sym[:, 0] = np.arange(n) % 50 + 1
    sym[:, 1] = big
    sym[0, 1] = 3
    t0 = time.perf_counter()
    out = jpeg_amd.rle_decode_plane(torch.from_numpy(sym).to(gpu), nb)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 5.0
    want = np.full(n, 2, np.int16)
    want[:3] = 1
    assert np.array_equal(out.cpu().numpy().reshape(-1), want)


def test_lena_bin_through_device_rle(gpu, orc, lena_bgr):
    """config 1 end to end on the device up to the entropy coder: fused encode ->
    device RLE -> the oracle's Huffman/.bin writer -> the committed golden file."""
    import jpeg_amd
    qy, qc = jpeg_amd.quality_tables(50)
    coefs = jpeg_amd.encode_image(torch.from_numpy(lena_bgr).to(gpu), qy, qc)
    streams = [jpeg_amd.rle_encode_plane(c)[0].cpu().numpy().reshape(-1) for c in coefs]
    data = orc.write_bin(*streams, 1024, 1024)
    gold = open(os.path.join(GOLD, "compressed_image_cpu_50.bin"), "rb").read()
    assert data == gold and hashlib.sha256(data).hexdigest().startswith("7870d3fec500635c")
    # and back: the golden file's symbol streams through the device decodeRLE + fused decode
    # (loadEncodedData expands the pad bits too, encode_decode.cpp:1261-1263: the
    # streams may carry spurious trailing symbols, possibly an odd number of them)
    dims, streams = orc.read_bin(gold)
    assert list(dims) == [1024, 1024, 512, 512, 512, 512]
    planes = [jpeg_amd.rle_decode_plane(torch.from_numpy(s[:s.size // 2 * 2].reshape(-1, 2).copy()).to(gpu), n)
              for s, n in zip(streams, (128 * 128, 64 * 64, 64 * 64))]
    for p, c in zip(planes, coefs):
        assert torch.equal(p, c)


def test_rle_full_size_round_trip(gpu):
    """8192^2 luma plane (1 Mi blocks): decode(encode(c)) == c, totals consistent."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    qy, qc = jpeg_amd.quality_tables(50)
    img = synth_image(8192, 8192, 0x8192, "photo", device=gpu)
    cy, ccb, ccr = jpeg_amd.encode_image(img, qy, qc)
    for c in (cy, ccb):
        pairs, offs = jpeg_amd.rle_encode_plane(c)
        assert int(offs[-1]) == pairs.shape[0] and int(pairs[:, 1].sum()) == c.numel()
        assert torch.equal(jpeg_amd.rle_decode_plane(pairs, c.shape[0]), c)
