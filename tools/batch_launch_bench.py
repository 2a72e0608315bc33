"""Developer helper: 64 images of 1024^2 (the reference's Lena size) per step, encoded and decoded
image by image (128 launches) and as one batch = one tall image (2 launches)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, jpeg_amd
from jpeg_amd.synth import synth_image
jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
n, size = 64, 1024
batch = torch.stack([synth_image(size, size, 500 + i, "photo", device=dev) for i in range(n)])
def one_by_one():
    for i in range(n):
        c = jpeg_amd.encode_image(batch[i], qy, qc, out=outs[i])
        jpeg_amd.decode_image(*c, size, size, qy, qc, out=recs[i])
def batched():
    c = jpeg_amd.encode_image(batch.view(n * size, size, 3), qy, qc, out=bout)
    jpeg_amd.decode_image(*c, n * size, size, qy, qc, out=brec)
outs = [tuple(torch.empty((k, 64), dtype=torch.int16, device=dev) for k in (size * size // 64, size * size // 256, size * size // 256)) for _ in range(n)]
recs = [torch.empty((size, size, 3), dtype=torch.uint8, device=dev) for _ in range(n)]
bout = tuple(torch.empty((n * k, 64), dtype=torch.int16, device=dev) for k in (size * size // 64, size * size // 256, size * size // 256))
brec = torch.empty((n * size, size, 3), dtype=torch.uint8, device=dev)
for name, fn in (("image by image (128 launches)", one_by_one), ("one batch (2 launches)", batched)):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    print(json.dumps({"how": name, "ms_per_64_images": best * 1e3, "mpixel_s": n * size * size / best / 1e6}))
assert torch.equal(brec.view(n, size, size, 3)[5], recs[5])
