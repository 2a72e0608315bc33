"""The entropy stage with both passes over the symbols on the device (entropy_kernels.hip,
jpegamd_bin_encode_dev): the container must equal the host coder's byte for
byte, which the CPU suite pins to the reference's files and to the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _streams(rng, nb, p_val=0.45, p_run=0.3, spread=1):
    n = int(nb * 4.3)
    vals = (rng.geometric(p_val, n) - 1) * rng.choice([-1, 1], n) * spread
    runs = np.minimum(rng.geometric(p_run, n), 64)
    return np.stack([vals, runs], 1).astype(np.int32).reshape(-1)


@pytest.mark.parametrize("seed,nby,spread", [(1, 64, 1), (2, 4096, 1), (3, 70001, 37), (4, 262144, 1)])
def test_device_container_equals_host_container(gpu, orc, seed, nby, spread):
    """... and the oracle's own writer (real std::unordered_map / std::priority_queue, the
    reference's string of '0'/'1' characters) on the same streams: not only a self-comparison"""
    import jpeg_amd
    rng = np.random.default_rng(seed)
    sy, scb, scr = _streams(rng, nby, spread=spread), _streams(rng, nby // 4 + 1, 0.7), _streams(rng, nby // 4 + 1, 0.2, 0.6)
    want = jpeg_amd.bin_encode(sy, scb, scr, 128, 256)
    got = jpeg_amd.bin_encode_dev(*(torch.from_numpy(x).to(gpu) for x in (sy, scb, scr)), 128, 256)
    assert got == want
    if nby <= 70001:  # (the oracle builds the bit string character by character)
        assert got == orc.write_bin(sy, scb, scr, 128, 256)


def test_edge_streams(gpu):
    """empty planes, two-symbol planes (gray chroma: every block is (0, 64)), a single repeated
    symbol (the empty code), ragged lengths that end inside a 32-bit word"""
    import jpeg_amd
    dev = lambda a: torch.tensor(a, dtype=torch.int32, device=gpu)
    cases = [
        ([], [], []),
        ([5, 1, 0, 63], [0, 64] * 100, [0, 64]),
        ([7, 7, 7, 7, 7, 7], [1, 2, 3], [9]),           # one distinct symbol -> zero payload bits
        (list(range(-40, 41)) * 3 + [1], [0, 64] * 3333, [2, 1, -2, 1] * 17 + [3]),
    ]
    for y, cb, cr in cases:
        want = jpeg_amd.bin_encode(np.array(y, np.int32), np.array(cb, np.int32), np.array(cr, np.int32), 16, 16)
        assert jpeg_amd.bin_encode_dev(dev(y), dev(cb), dev(cr), 16, 16) == want


def test_inputs_beyond_the_device_tables_fall_back_to_the_host_coder(gpu, monkeypatch):
    import jpeg_amd
    rng = np.random.default_rng(9)
    sy = _streams(rng, 3000)
    sy[100] = 70000   # a symbol outside int16 (never produced by the kernels)
    scb, scr = _streams(rng, 700), _streams(rng, 700)
    want = jpeg_amd.bin_encode(sy, scb, scr, 64, 64)
    assert jpeg_amd.bin_encode_dev(*(torch.from_numpy(x).to(gpu) for x in (sy, scb, scr)), 64, 64) == want
    # Fibonacci frequencies give codes longer than 32 bits: host path again
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    long_codes = np.concatenate([np.full(f, k, np.int32) for k, f in enumerate(fib[:36])])
    rng.shuffle(long_codes)
    want = jpeg_amd.bin_encode(long_codes, scb, scr, 64, 64)
    assert jpeg_amd.bin_encode_dev(torch.from_numpy(long_codes).to(gpu), torch.from_numpy(scb).to(gpu),
                                   torch.from_numpy(scr).to(gpu), 64, 64) == want
    monkeypatch.setenv("JPEGAMD_ENTROPY", "host")
    assert jpeg_amd.bin_encode_dev(*(torch.from_numpy(x).to(gpu) for x in (sy, scb, scr)), 64, 64) == \
        jpeg_amd.bin_encode(sy, scb, scr, 64, 64)


def test_compress_image_same_bytes_with_either_entropy_stage(gpu, lena_bgr, monkeypatch):
    import jpeg_amd
    from jpeg_amd.synth import synth_image
    for img, qf in ((lena_bgr, 50), (synth_image(528, 1040, 5, "noise").numpy(), 100),
                    (synth_image(64, 512, 6, "gray").numpy(), 90)):
        qy, qc = jpeg_amd.quality_tables(qf)
        monkeypatch.delenv("JPEGAMD_ENTROPY", raising=False)
        on_device = jpeg_amd.compress_image(img, qy, qc)
        monkeypatch.setenv("JPEGAMD_ENTROPY", "host")
        assert jpeg_amd.compress_image(img, qy, qc) == on_device
