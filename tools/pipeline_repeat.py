"""Developer helper: image -> .bin -> image from host memory, several calls in one process
(the first call also creates the staging buffers and streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if len(sys.argv) > 2:
    from jpeg_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
import jpeg_amd
from jpeg_amd.synth import synth_image
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
jpeg_amd.init(0)
qy, qc = jpeg_amd.quality_tables(50)
himg = synth_image(size, size, 0x8192, "photo", device="cuda:0").cpu().numpy()
for i in range(6):
    t0 = time.perf_counter(); data = jpeg_amd.compress_image(himg, qy, qc); t1 = time.perf_counter()
    rec = jpeg_amd.decompress_image(data, qy, qc); t2 = time.perf_counter()
    print(f"call {i}: image -> .bin {1e3 * (t1 - t0):6.2f} ms   .bin -> image {1e3 * (t2 - t1):6.2f} ms   ({len(data)} bytes)")
with jpeg_amd.Context(0) as ctx:  # the same through a context: device buffers kept between calls, output array reused
    out = np.empty_like(himg)
    for i in range(6):
        t0 = time.perf_counter(); d2 = ctx.compress_image(himg, qy, qc); t1 = time.perf_counter()
        ctx.decompress_image(d2, qy, qc, out=out); t2 = time.perf_counter()
        print(f"context call {i}: image -> .bin {1e3 * (t1 - t0):6.2f} ms   .bin -> image {1e3 * (t2 - t1):6.2f} ms")
    assert d2 == data and np.array_equal(out, rec)
