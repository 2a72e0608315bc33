"""Developer helper: n images per step (default 64 of 1024^2, the reference's Lena size; SURVEY 8d
config 4 is `64 2160 3840`), encoded and decoded image by image (2n launches) and as one batch =
one tall image (2 launches).  python tools/batch_launch_bench.py [n rows cols]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, jpeg_amd
from jpeg_amd.synth import synth_image
jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
n, rows, cols = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 1024, 1024)
batch = torch.stack([synth_image(rows, cols, 500 + i, "photo", device=dev) for i in range(n)])
def one_by_one():
    for i in range(n):
        c = jpeg_amd.encode_image(batch[i], qy, qc, out=outs[i])
        jpeg_amd.decode_image(*c, rows, cols, qy, qc, out=recs[i])
def batched():
    c = jpeg_amd.encode_image(batch.view(n * rows, cols, 3), qy, qc, out=bout)
    jpeg_amd.decode_image(*c, n * rows, cols, qy, qc, out=brec)
outs = [tuple(torch.empty((k, 64), dtype=torch.int16, device=dev) for k in (rows * cols // 64, rows * cols // 256, rows * cols // 256)) for _ in range(n)]
recs = [torch.empty((rows, cols, 3), dtype=torch.uint8, device=dev) for _ in range(n)]
bout = tuple(torch.empty((n * k, 64), dtype=torch.int16, device=dev) for k in (rows * cols // 64, rows * cols // 256, rows * cols // 256))
brec = torch.empty((n * rows, cols, 3), dtype=torch.uint8, device=dev)
for name, fn in ((f"image by image ({2 * n} launches)", one_by_one), ("one batch (2 launches)", batched)):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    print(json.dumps({"how": name, "images": f"{n} x {rows}x{cols}", "ms_per_step": best * 1e3, "mpixel_s": n * rows * cols / best / 1e6}))
assert torch.equal(brec.view(n, rows, cols, 3)[5], recs[5])
