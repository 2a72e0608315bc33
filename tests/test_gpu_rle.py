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
