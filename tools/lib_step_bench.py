"""Developer helper: in-sequence step time (encode, decode, encode, ...) and the kernels' own
durations for an alternative build of the library: python tools/lib_step_bench.py <lib.so>."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jpeg_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import jpeg_amd
from jpeg_amd.synth import synth_image
size = 8192
jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
bgr = synth_image(size, size, 0x8192, "photo", device=dev)
coefs = jpeg_amd.encode_image(bgr, qy, qc)
rec = torch.empty_like(bgr)
def step():
    jpeg_amd.encode_image(bgr, qy, qc, out=coefs)
    jpeg_amd.decode_image(*coefs, size, size, qy, qc, out=rec)
for _ in range(5):
    step()
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 50)
e, d = jpeg_amd.time_image(bgr, coefs, rec, qy, qc, 30)
print(json.dumps({"lib": sys.argv[1], "transform": os.environ.get("JPEGAMD_TRANSFORM", "valu"),
                  "step_us": best * 1e6, "enc_us": e * 1e3, "dec_us": d * 1e3}))
