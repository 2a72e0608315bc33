"""Pins the CPU oracle (oracle/jpeg_oracle.c, jpeg_oracle_entropy.cpp) to the
reference: its committed goldens, its Wikipedia-block walk-through, and vectors
produced by the reference's own jpeg_function.cpp (oracle/_ref).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _planes(orc, bgr):
    return orc.chroma_subsample(orc.rgb2ycbcr(bgr))


@pytest.mark.parametrize("flavour,name,sha", [
    (0, "cpu", "7870d3fec500635c7ef9dc5a98f985f31dbb2c91b0beb8d0c6ba9f9dbf82b2bf"),
    (1, "omp", "4d144b5c46e5aa0a380348addaa4fff54878737bb0f1efe87bb3e0c70cb47fb9"),
])
def test_lena_bin_and_pixels_match_committed_goldens(orc, lena_bgr, flavour, name, sha):
    """config 1: lena_color_1024.png, QF=50, reference CPU (and OMP) path."""
    from PIL import Image
    qy, qc = orc.quality_tables(50)
    y, cb, cr = _planes(orc, lena_bgr)
    coefs, syms = zip(*(orc.encode_plane(p, q, flavour) for p, q in ((y, qy), (cb, qc), (cr, qc))))
    data = orc.write_bin(*syms, 1024, 1024)
    gold = open(os.path.join(GOLD, f"compressed_image_{name}_50.bin"), "rb").read()
    assert data == gold
    assert hashlib.sha256(data).hexdigest() == sha
    # header facts recorded in SURVEY.md 8c
    hdr = np.frombuffer(data[:36], np.int32)
    assert hdr[:6].tolist() == [1024, 1024, 512, 512, 512, 512]
    if flavour == 0:
        assert hdr[6:9].tolist() == [106240, 11698, 10575]
    # decode side: loader + Huffman decode + decodeRLE + inverse chain
    dims, dec = orc.read_bin(gold)
    assert dims.tolist() == [1024, 1024, 512, 512, 512, 512]
    planes = []
    for s, c, shape, q in zip(dec, coefs, ((1024, 1024), (512, 512), (512, 512)), (qy, qc, qc)):
        nb = (shape[0] // 8) * (shape[1] // 8)
        back, done = orc.decode_rle(s, nb)
        assert done == nb and np.array_equal(back, c)
        planes.append(orc.decode_plane(back, shape[0], shape[1], q, flavour))
    rec = orc.decode_image(*coefs, 1024, 1024, qy, qc, flavour)
    gold_px = np.array(Image.open(os.path.join(GOLD, f"decompress_image_{name}_50.png")).convert("RGB"))[:, :, ::-1]
    assert np.array_equal(rec, gold_px)
    mse = orc.mse(lena_bgr, rec)
    want = {0: (19.4944, 35.2317), 1: (19.5247, 35.2250)}[flavour]
    assert abs(mse - want[0]) < 5e-4 and abs(orc.psnr(mse) - want[1]) < 5e-4


def test_wikipedia_block_known_answers(orc):
    kat = json.load(open(os.path.join(GOLD, "wikipedia_block.json")))
    px = np.array(kat["pixels"], np.int32)
    q = np.array(kat["q"], np.int32)
    assert np.array_equal(q, orc.quality_tables(50)[0])
    d = orc.dct_block(px - 128)
    assert np.allclose(d, kat["dct"], atol=1e-4)
    # values printed in SURVEY.md 8c / docs/JPEG_Report.pdf 3.7-3.9
    assert [round(float(x), 2) for x in d[:8]] == [-415.37, -30.19, -61.2, 27.24, 56.12, -20.1, -2.39, 0.46]
    c = orc.quantize_block(d, q)
    assert c.astype(int).tolist() == kat["quantised"]
    assert c[:8].astype(int).tolist() == [-26, -3, -6, 2, 2, -1, 0, 0]
    zz = orc.zigzag_block(c)
    assert zz.tolist() == kat["zigzag"]
    assert orc.rle_block(zz).tolist() == kat["rle"]
    back = orc.idct_block(c * q.astype(np.float32)) + 128
    assert back.tolist() == kat["restored"]
    assert back[:8].tolist() == [62, 65, 57, 60, 72, 63, 60, 82]


def test_reference_block_vectors(orc):
    """400 blocks pushed through the reference's own dct/quantize/zigzag/RLE/idct."""
    v = np.load(os.path.join(GOLD, "ref_block_vectors.npz"))
    stream = []
    for i in range(v["pixels"].shape[0]):
        q = orc.quality_tables(int(v["quality"][i]))[0]
        d = orc.dct_block(v["pixels"][i] - 128)
        assert np.array_equal(d, v["dct"][i])
        c = orc.quantize_block(d, q)
        assert np.array_equal(c, v["quantised"][i])
        assert not np.signbit(c[c == 0]).any()
        zz = orc.zigzag_block(c)
        assert np.array_equal(zz, v["zigzag"][i])
        r = orc.rle_block(zz)
        assert len(r) == v["rle_len"][i]
        stream.append(r)
        assert np.array_equal(orc.idct_block(c * q.astype(np.float32)) + 128, v["restored"][i])
    stream = np.concatenate(stream)
    assert np.array_equal(stream, v["rle"])
    # entropy coder: same dictionary order and same bit string as the reference
    data = orc.write_bin(stream, stream[:2], stream[:2], 16, 16)
    ylen = int(np.frombuffer(data[24:28], np.int32)[0])
    bits = np.unpackbits(np.frombuffer(data[36:36 + ylen], np.uint8))[:v["huffman_bits"].size]
    assert np.array_equal(bits + ord("0"), v["huffman_bits"])
    dict_off = 36 + ylen + int(np.frombuffer(data[28:32], np.int32)[0]) + int(np.frombuffer(data[32:36], np.int32)[0])
    n = int(np.frombuffer(data[dict_off:dict_off + 4], np.int32)[0])
    pairs = np.frombuffer(data[dict_off + 12:dict_off + 12 + 8 * n], np.int32).reshape(n, 2)
    assert np.array_equal(pairs, v["freq_dict_order"])


@pytest.mark.skipif(not os.path.exists(oracle_lib.REF_SO), reason="oracle/_ref not built (no reference checkout)")
def test_oracle_equals_live_reference_on_random_blocks(orc):
    ref = oracle_lib.Reference()
    rng = np.random.default_rng(5)
    for i in range(200):
        px = rng.integers(0, 256, 64).astype(np.int32) if i % 3 else np.full(64, rng.integers(0, 256), np.int32)
        qf = int(rng.integers(1, 101))
        q = ref.luma_table(qf)
        assert np.array_equal(q, orc.quality_tables(qf)[0])
        d = ref.recenter_dct(px)
        assert np.array_equal(d, orc.dct_block(px - 128))
        c = ref.quantize(d, q)
        assert np.array_equal(c, orc.quantize_block(d, q))
        zz = ref.zigzag(c)
        assert np.array_equal(ref.decode_block(zz, q), orc.idct_block(c * q.astype(np.float32)) + 128)


def test_zigzag_table_is_the_reference_walk(orc):
    assert orc.lib.orc_zigzag_selfcheck() == 1
    z = (oracle_lib.C.c_uint8 * 64).in_dll(orc.lib, "orc_zigzag_order")
    assert list(z) == oracle_lib.ZIGZAG.tolist()
    assert sorted(z) == list(range(64))


def test_quality_tables(orc):
    import jpeg_amd
    for qf in range(1, 101):
        a, b = orc.quality_tables(qf)
        c, d = jpeg_amd.quality_tables(qf)  # the product's host-side mirror of main.cpp:51-80
        assert np.array_equal(a, c) and np.array_equal(b, d)
        assert a.min() >= 1 and a.max() <= 255
    assert (orc.quality_tables(100)[0] == 1).all() and (orc.quality_tables(1)[0] == 255).all()
    for qf in (50, 89, 90, 100):
        a, b = orc.analysis_tables(qf)
        c, d = jpeg_amd.analysis_tables(qf)
        assert np.array_equal(a, c) and np.array_equal(b, d)
    assert orc.analysis_tables(95)[0][5 * 8 + 5] == 12  # analysis.cpp:51
    with pytest.raises(ValueError):
        jpeg_amd.quality_tables(0)


def test_rle_edge_cases(orc):
    z = np.zeros(64, np.int32)
    assert orc.rle_block(z).tolist() == [0, 64]
    z = np.arange(64, dtype=np.int32)
    r = orc.rle_block(z)
    assert len(r) == 128 and r[1::2].tolist() == [1] * 64
    coef, done = orc.decode_rle(np.array([5, 64, 7, 10], np.int32), 2)
    assert done == 1 and (coef[0] == 5).all()  # short trailing block is not a block (encode_decode.cpp:702)
    coef, done = orc.decode_rle(np.array([1, 100, 2, 28, 9, 64], np.int32), 2)
    assert done == 2 and (coef[0] == 1).all() and coef[1, :36].tolist() == [1] * 36 and (coef[1, 36:] == 2).all()
