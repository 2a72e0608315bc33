"""jpegamd_rle_encode_plane_noalloc_dev: the run-length encode entry point that allocates
nothing (its per-workgroup pair counts live in a fixed device array inside the library).
Checked through the C ABI against jpegamd_rle_encode_plane_dev (the workspace form, same
kernels), bit for bit, on two calls in a row, the second with a smaller input -- the array
keeps the first call's contents.  Collected last (file name): a failure here must not hide
the other GPU tests under `pytest -x`."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _encode_both(lib, coef, stream):
    from jpeg_amd import _lib
    nb = coef.shape[0]
    dev = coef.device
    cap = nb * 64
    outs = []
    for noalloc in (False, True):
        pairs = torch.full((cap, 2), -12345, dtype=torch.int32, device=dev)
        offs = torch.full((nb + 1,), -1, dtype=torch.int32, device=dev)
        if noalloc:
            _lib.check(lib.jpegamd_rle_encode_plane_noalloc_dev(coef.data_ptr(), nb, pairs.data_ptr(), cap,
                                                                offs.data_ptr(), stream))
        else:
            work = torch.empty(lib.jpegamd_rle_workspace_bytes(nb), dtype=torch.uint8, device=dev)
            _lib.check(lib.jpegamd_rle_encode_plane_dev(coef.data_ptr(), nb, pairs.data_ptr(), cap, offs.data_ptr(),
                                                        work.data_ptr(), work.numel(), stream))
        torch.cuda.synchronize()
        outs.append((pairs.cpu().numpy(), offs.cpu().numpy()))
    return outs


def test_noalloc_rle_encode_equals_workspace_form_twice_in_a_row(gpu, orc):
    from jpeg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(99)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # first a large plane (many workgroups of scratch in use), then a smaller one
    for nb in (70001, 513, 1):
        coef_h = np.zeros((nb, 64), np.int16)
        m = rng.random((nb, 64)) < 0.2
        coef_h[m] = rng.integers(-40, 41, size=int(m.sum()), dtype=np.int16)
        coef = torch.from_numpy(coef_h).to(gpu)
        (want_pairs, want_offs), (got_pairs, got_offs) = _encode_both(lib, coef, stream)
        total = int(want_offs[nb])
        assert int(got_offs[nb]) == total
        assert np.array_equal(got_offs, want_offs), nb
        assert np.array_equal(got_pairs[:total], want_pairs[:total]), nb
        assert (got_pairs[total:] == -12345).all()          # nothing written beyond the stream
        assert np.array_equal(got_pairs[:total].reshape(-1), orc.rle_stream(coef_h)), nb


def test_noalloc_rle_encode_rejects_planes_beyond_its_scratch(gpu):
    from jpeg_amd import _lib, JpegAmdError
    lib = _lib.load()
    limit = lib.jpegamd_rle_noalloc_max_blocks()
    assert limit == 1 << 22
    dummy = torch.zeros(64, dtype=torch.int16, device=gpu)
    out = torch.zeros(16, dtype=torch.int32, device=gpu)
    with pytest.raises(JpegAmdError, match="exceeds the built-in scratch"):
        _lib.check(lib.jpegamd_rle_encode_plane_noalloc_dev(dummy.data_ptr(), limit + 1, out.data_ptr(), 8,
                                                            out.data_ptr(), None))
