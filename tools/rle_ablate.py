"""Developer helper: time the run-length stage (three planes, generic path) at 8192^2."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, jpeg_amd
from jpeg_amd import _lib
from jpeg_amd.synth import synth_image
jpeg_amd.init(0)
lib = _lib.load()
dev = torch.device("cuda:0")
size = 8192
qy, qc = jpeg_amd.quality_tables(50)
bgr = synth_image(size, size, 0x8192, "photo", device=dev)
coefs = jpeg_amd.encode_image(bgr, qy, qc)
nbs = [int(c.shape[0]) for c in coefs]
work = torch.empty(lib.jpegamd_rle_workspace_bytes(sum(nbs)), dtype=torch.uint8, device=dev)
offs = [torch.empty(nb + 1, dtype=torch.int32, device=dev) for nb in nbs]
caps = [nb * 16 for nb in nbs]
pairs = [torch.empty((cap, 2), dtype=torch.int32, device=dev) for cap in caps]
arr = lambda vals, t: (t * 3)(*vals)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
args = (arr([c.data_ptr() for c in coefs], ctypes.c_void_p), arr(nbs, ctypes.c_size_t),
        arr([p.data_ptr() for p in pairs], ctypes.c_void_p), arr(caps, ctypes.c_size_t),
        arr([o.data_ptr() for o in offs], ctypes.c_void_p), work.data_ptr(), work.numel(), stream)
for _ in range(3):
    _lib.check(lib.jpegamd_rle_encode_image_dev(*args))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    _lib.check(lib.jpegamd_rle_encode_image_dev(*args))
e1.record()
torch.cuda.synchronize()
print("rle ms per call", e0.elapsed_time(e1) / 20, flush=True)
