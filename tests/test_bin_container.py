"""The product's host entropy coder and .bin container (jpeg_amd/csrc/host/bin_container.cpp,
SURVEY.md 8f-1/8f-2) against the committed golden files and the oracle.  Host logic
only: no GPU is involved, so these run in the CPU suite."""
import hashlib
import os

import numpy as np
import pytest

import jpeg_amd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("flavour,name,sha", [
    (0, "cpu", "7870d3fec500635c7ef9dc5a98f985f31dbb2c91b0beb8d0c6ba9f9dbf82b2bf"),
    (1, "omp", "4d144b5c46e5aa0a380348addaa4fff54878737bb0f1efe87bb3e0c70cb47fb9"),
])
def test_golden_bin_is_reproduced_byte_for_byte(orc, lena_bgr, flavour, name, sha):
    qy, qc = orc.quality_tables(50)
    planes = orc.chroma_subsample(orc.rgb2ycbcr(lena_bgr))
    syms = [orc.encode_plane(p, q, flavour)[1] for p, q in zip(planes, (qy, qc, qc))]
    data = jpeg_amd.bin_encode(*syms, 1024, 1024)
    gold = open(os.path.join(GOLD, f"compressed_image_{name}_50.bin"), "rb").read()
    assert data == gold and hashlib.sha256(data).hexdigest() == sha
    # and read back: every symbol the reference's loader would produce, pad-bit symbols included
    dims, dec = jpeg_amd.bin_decode(gold)
    odims, odec = orc.read_bin(gold)
    assert dims.tolist() == odims.tolist() == [1024, 1024, 512, 512, 512, 512]
    for a, b, s in zip(dec, odec, syms):
        assert np.array_equal(a, b)
        assert np.array_equal(a[:s.size], s) and a.size - s.size <= 7


def _streams(rng):
    yield "geometric", rng.geometric(0.3, 5000).astype(np.int32) - 3
    yield "many equal frequencies", np.tile(np.arange(-20, 21, dtype=np.int32), 7)
    yield "two symbols", np.array([0, 64] * 300, np.int32)
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    yield "skewed (deep tree)", np.concatenate([np.full(f, i - 5, np.int32) for i, f in enumerate(fib)])
    yield "wide alphabet", rng.integers(-1024, 1025, 20000).astype(np.int32)
    yield "empty", np.empty(0, np.int32)
    yield "outside the dense window", np.array([100000, -70000, 5, 5, 100000, 3, 64], np.int32)


def test_random_streams_equal_the_oracle(orc):
    rng = np.random.default_rng(5)
    streams = list(_streams(rng))
    for i, (name, s) in enumerate(streams):
        others = [streams[(i + 1) % len(streams)][1], streams[(i + 2) % len(streams)][1]]
        if any(np.unique(x).size == 1 for x in (s, *others)):
            continue
        data = jpeg_amd.bin_encode(s, *others, 48, 32)
        assert data == orc.write_bin(s, *others, 48, 32), name
        if all(x.size for x in (s, *others)):
            dims, dec = jpeg_amd.bin_decode(data)
            assert dims.tolist() == [48, 32, 24, 16, 24, 16]
            for a, b in zip(dec, (s, *others)):
                assert np.array_equal(a[:b.size], b) and a.size - b.size <= 7, name


def test_large_planes_use_worker_threads_and_still_match(orc):
    """Above about half a million symbols per plane the histogram and the bit packing
    run on several threads (chunk-wise first-occurrence lists merged in order, chunks
    packed from their own bit offsets): the bytes must not change."""
    rng = np.random.default_rng(11)
    n = 3_300_001
    s = (rng.geometric(0.25, n).astype(np.int32) - 2) * rng.choice(np.array([-1, 1], np.int32), n)
    s[2_900_000:2_900_050] = np.arange(5000, 5050, dtype=np.int32)   # symbols whose first occurrence is late
    s[-1] = -777
    small = np.array([0, 64, 1, 1], np.int32)
    data = jpeg_amd.bin_encode(s, small, s[:1_100_003], 64, 64)
    assert data == orc.write_bin(s, small, s[:1_100_003], 64, 64)
    dims, dec = jpeg_amd.bin_decode(data)
    assert np.array_equal(dec[0][:n], s) and dec[0].size - n <= 7


@pytest.mark.parametrize("chunk_bytes", [8, 24, 100, 4096])
def test_decoding_in_chunks_gives_the_serial_symbols(orc, monkeypatch, chunk_bytes):
    """A plane's bit stream has no restart points; large planes are still decoded by several
    threads, each starting mid-stream and falling into step with the true decoding
    (PlaneDecoder).  JPEGAMD_DECODE_CHUNK_BYTES cuts small files into many chunks: every
    stream, the goldens and a long-code stream must decode to exactly the serial symbols,
    spurious pad-bit symbols included."""
    rng = np.random.default_rng(17)
    files = [open(os.path.join(GOLD, f"compressed_image_{n}_50.bin"), "rb").read() for n in ("cpu", "omp")]
    streams = [s for _, s in _streams(rng) if s.size and np.unique(s).size > 1]
    for i, s in enumerate(streams):
        files.append(jpeg_amd.bin_encode(s, streams[(i + 1) % len(streams)], streams[(i + 2) % len(streams)], 48, 32))
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    deep = np.concatenate([np.full(min(f, 400), i, np.int32) for i, f in enumerate(fib)])
    rng.shuffle(deep)
    files.append(jpeg_amd.bin_encode(deep, deep[::-1].copy(), deep[:500].copy(), 16, 16))
    # eight equally frequent symbols get eight 3-bit codes: a thread that starts off a multiple
    # of 3 NEVER falls into step, and its chunk is decoded as part of its predecessor's region
    # (several lengths: the chunk size depends on the machine's core count, and a chunk size
    # that happens to be a multiple of 3 bits would start every thread in step)
    for reps in (700, 701, 703, 706, 710, 715):
        fixed = rng.permutation(np.tile(np.arange(8, dtype=np.int32), reps))
        files.append(jpeg_amd.bin_encode(fixed, fixed[:808].copy(), fixed[:1608].copy(), 16, 16))
    monkeypatch.delenv("JPEGAMD_DECODE_CHUNK_BYTES", raising=False)
    serial = [jpeg_amd.bin_decode(f) for f in files]
    monkeypatch.setenv("JPEGAMD_DECODE_CHUNK_BYTES", str(chunk_bytes))
    for f, (dims, want) in zip(files, serial):
        odims, got = jpeg_amd.bin_decode(f)
        assert odims.tolist() == dims.tolist()
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    # and against the oracle's bit-serial loader on the goldens
    for f in files[:2]:
        _, odec = orc.read_bin(f)
        _, got = jpeg_amd.bin_decode(f)
        assert all(np.array_equal(a, b) for a, b in zip(got, odec))


def test_very_long_codes_take_the_bitwise_path(orc):
    fib = [1, 1]
    while len(fib) < 62:   # depth 61 > 56 bits: beyond the flat code table
        fib.append(fib[-1] + fib[-2])
    # frequencies that large cannot be materialised; scale the tail down but keep the chain shape
    f = np.minimum(np.array(fib, dtype=np.int64), 3000)
    f = np.maximum.accumulate(f)
    f[:12] = [1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144]
    s = np.concatenate([np.full(int(n), i, np.int32) for i, n in enumerate(f)])
    np.random.default_rng(1).shuffle(s)
    small = np.array([1, 2, 1, 2, 3], np.int32)
    assert jpeg_amd.bin_encode(s, small, small, 16, 16) == orc.write_bin(s, small, small, 16, 16)


def test_wikipedia_block_huffman_string():
    """the 107-bit code string of the walk-through block."""
    rle = np.array([-26, 1, -3, 1, 0, 1, -3, 1, -2, 1, -6, 1, 2, 1, -4, 1, 1, 1, -3, 1, 1, 2, 5, 1, 1, 1, 2, 1,
                    -1, 1, 1, 1, -1, 1, 2, 1, 0, 5, -1, 2, 0, 38], np.int32)
    want = ("1000001100011010110001111001001001010100010001100001011111100010101110000111001010110111111111"
            "0101110110011")
    two = np.array([0, 64], np.int32)
    data = jpeg_amd.bin_encode(rle, two, two, 8, 8)
    hdr = np.frombuffer(data[:36], np.int32)
    assert hdr[6] == (len(want) + 7) // 8
    bits = "".join(f"{b:08b}" for b in data[36:36 + hdr[6]])
    assert bits[:len(want)] == want and set(bits[len(want):]) <= {"0"}


def test_malformed_files_fail_loudly():
    gold = open(os.path.join(GOLD, "compressed_image_cpu_50.bin"), "rb").read()
    for cut in (10, 36, 5000, len(gold) - 4):
        with pytest.raises(jpeg_amd.JpegAmdError):
            jpeg_amd.bin_decode(gold[:cut])
    one = np.array([7, 7, 7], np.int32)            # a one-symbol plane encodes to nothing ...
    two = np.array([0, 64], np.int32)
    data = jpeg_amd.bin_encode(one, two, two, 16, 16)
    assert np.frombuffer(data[:36], np.int32)[6] == 0
    dims, dec = jpeg_amd.bin_decode(data)            # ... and decodes to nothing (empty payload)
    assert dec[0].size == 0
    with pytest.raises(jpeg_amd.JpegAmdError, match="99999"):
        jpeg_amd.bin_encode(np.array([99999, 1], np.int32), two, two, 16, 16)
