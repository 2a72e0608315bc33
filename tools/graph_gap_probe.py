"""Developer probe: per-step time of encode->decode when launched one by one, as a HIP graph
of one step, and as a HIP graph of ten steps (what do the kernel-to-kernel gaps cost?)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jpeg_amd
from jpeg_amd import _lib
from jpeg_amd.synth import synth_image

jpeg_amd.init(0)
dev = torch.device("cuda:0")
rows = cols = 8192
qy, qc = jpeg_amd.quality_tables(50)
bgr = synth_image(rows, cols, 0x8192, "photo", device=dev)
coefs = jpeg_amd.encode_image(bgr, qy, qc)
rec = torch.empty_like(bgr)
lib = _lib.load()
ql32, qc32 = (np.ascontiguousarray(np.asarray(t, np.int32).reshape(64)) for t in (qy, qc))

def launch(stream):
    s = ctypes.c_void_p(stream.cuda_stream)
    _lib.check(lib.jpegamd_encode_image_dev(bgr.data_ptr(), rows, cols, cols * 3, ql32.ctypes.data, qc32.ctypes.data,
                                            coefs[0].data_ptr(), coefs[1].data_ptr(), coefs[2].data_ptr(), s))
    _lib.check(lib.jpegamd_decode_image_dev(coefs[0].data_ptr(), coefs[1].data_ptr(), coefs[2].data_ptr(), rows, cols,
                                            ql32.ctypes.data, qc32.ctypes.data, rec.data_ptr(), cols * 3, s))

def timed(fn, steps_per_call, calls):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (calls * steps_per_call) * 1e6

side = torch.cuda.Stream()
print("one by one      : %.1f us per step" % timed(lambda: launch(torch.cuda.current_stream()), 1, 100))
for n in (1, 10):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        launch(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n):
            launch(side)
    print("graph of %2d step%s: %.1f us per step" % (n, "s" if n > 1 else " ", timed(g.replay, n, 100 // n)))
