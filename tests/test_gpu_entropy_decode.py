"""huffman_decode on the device (entropy_decode_kernels.hip, jpegamd_bin_decode_dev; SURVEY 8 f-2,
encode_decode.cpp:468-482 + the loader's every-bit-of-every-byte quirk :1261-1263): the symbol
streams must be the host decoder's, count and pad-bit symbols included.  The host decoder is
pinned by the CPU suite to the oracle's bit-serial loader and the reference's committed files;
the goldens are compared with the oracle directly here as well."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(data, gpu):
    import jpeg_amd
    dims, want = jpeg_amd.bin_decode(data)
    ddims, got = jpeg_amd.bin_decode_dev(data, gpu)
    assert ddims.tolist() == dims.tolist()
    for k, (a, b) in enumerate(zip(got, want)):
        a = a.cpu().numpy()
        assert a.size == b.size, (k, a.size, b.size)
        assert np.array_equal(a, b), (k, int((a != b).argmax()))
    return want


def test_goldens_decode_to_the_oracle_loaders_symbols(gpu, orc):
    import jpeg_amd
    for name in ("cpu", "omp"):
        data = open(os.path.join(GOLD, f"compressed_image_{name}_50.bin"), "rb").read()
        _same(data, gpu)
        _, odec = orc.read_bin(data)
        _, got = jpeg_amd.bin_decode_dev(data, gpu)
        assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(got, odec))


def test_stream_shapes(gpu):
    """short and long streams, two-symbol alphabets, deep trees whose codes outgrow the 12-bit
    window (bit by bit from the node reached), fixed-length codes (a thread that starts off a code
    boundary NEVER falls into step: its subsequence is absorbed by the chain through the
    workgroup), streams that end inside a byte, inside a word, and exactly on one"""
    import jpeg_amd
    rng = np.random.default_rng(17)
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    deep = np.concatenate([np.full(min(f, 400), i, np.int32) for i, f in enumerate(fib)])
    rng.shuffle(deep)
    streams = [
        rng.geometric(0.3, 5000).astype(np.int32) - 3,
        np.tile(np.arange(-20, 21, dtype=np.int32), 7),
        np.array([0, 64] * 300, np.int32),
        rng.integers(-1024, 1025, 20000).astype(np.int32),
        deep,
        np.array([1, 2, 1, 2, 3], np.int32),
        (rng.geometric(0.25, 300001).astype(np.int32) - 2) * rng.choice(np.array([-1, 1], np.int32), 300001),
    ]
    for i, s in enumerate(streams):
        data = jpeg_amd.bin_encode(s, streams[(i + 1) % len(streams)], streams[(i + 2) % len(streams)], 48, 32)
        want = _same(data, gpu)
        assert np.array_equal(want[0][:s.size], s) and want[0].size - s.size <= 7
    for reps in (700, 701, 703, 706, 710, 715, 9000, 9001):      # eight 3-bit codes
        fixed = rng.permutation(np.tile(np.arange(8, dtype=np.int32), reps))
        _same(jpeg_amd.bin_encode(fixed, fixed[:808].copy(), fixed[:1608].copy(), 16, 16), gpu)
    for n in range(1, 40):                                        # every tail length
        _same(jpeg_amd.bin_encode(streams[0][:n + 5], streams[3][:n * 7 + 2], streams[1][:n + 2], 16, 16), gpu)


def test_symbols_outside_int16_take_the_host_decoder(gpu):
    import jpeg_amd
    s = np.array([100000, -70000, 5, 5, 100000, 3, 64], np.int32)
    small = np.array([1, 2, 1, 2, 3], np.int32)
    _same(jpeg_amd.bin_encode(s, small, small, 16, 16), gpu)


def test_real_image_streams_and_the_whole_path(gpu):
    """8192^2 would be the bench image; 2048^2 keeps the host decoder's share of the test short.
    The device symbols feed decodeRLE and the fused decoder inside jpegamd_decompress_image: the
    image must equal the one decoded from the encoder's own coefficient planes."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    for kind, size, qf in (("photo", 2048, 50), ("noise", 512, 90), ("gray", 1024, 10)):
        qy, qc = jpeg_amd.quality_tables(qf)
        img = synth_image(size, size, 3, kind, device=gpu)
        data = jpeg_amd.compress_image(img.cpu().numpy(), qy, qc)
        _same(data, gpu)
        rec = jpeg_amd.decompress_image(data, qy, qc)
        want = jpeg_amd.decode_image(*jpeg_amd.encode_image(img, qy, qc), size, size, qy, qc).cpu().numpy()
        assert np.array_equal(rec, want)


def test_host_and_device_entropy_decoders_give_the_same_image(gpu, monkeypatch):
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    qy, qc = jpeg_amd.quality_tables(75)
    img = synth_image(528, 1040, 8, "photo").numpy()
    data = jpeg_amd.compress_image(img, qy, qc)
    dev = jpeg_amd.decompress_image(data, qy, qc)
    monkeypatch.setenv("JPEGAMD_ENTROPY", "host")
    assert np.array_equal(jpeg_amd.decompress_image(data, qy, qc), dev)


def test_full_size_round_trip_on_the_device(gpu):
    """BASELINE configs[1] size (8192^2, no host decoder at that size in seconds): fused encode + pairs ->
    jpegamd_bin_encode_dev -> jpegamd_bin_decode_dev must give back the pair streams symbol for symbol
    (at most seven pad-bit symbols behind each, encode_decode.cpp:1261-1263), and decodeRLE of them the
    encoder's coefficient planes.  Y has about 560 workgroups of 256 subsequences here: the emit kernel's
    sum of the workgroup totals before its own, and the payload windows in LDS at every word phase."""
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    size = 8192
    qy, qc = jpeg_amd.quality_tables(50)
    bgr = synth_image(size, size, 0x8192, "photo", device=gpu)
    coefs, streams = jpeg_amd.encode_image_rle(bgr, qy, qc)
    pairs = [p for p, _ in streams]
    data = jpeg_amd.bin_encode_dev(*[p.reshape(-1) for p in pairs], size, size)
    dims, syms = jpeg_amd.bin_decode_dev(data, gpu)
    assert dims.tolist() == [size, size, size // 2, size // 2, size // 2, size // 2]
    for k, (s, p, c) in enumerate(zip(syms, pairs, coefs)):
        n = p.numel()
        assert 0 <= s.numel() - n <= 7, (k, s.numel(), n)
        assert torch.equal(s[:n], p.reshape(-1)), k
        back = jpeg_amd.rle_decode_plane(s[:n].reshape(-1, 2), c.shape[0])
        assert torch.equal(back, c), k
