"""Developer helper: transform-only timing through the plane entry points."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, jpeg_amd
jpeg_amd.init(0)
dev = torch.device("cuda:0")
q = jpeg_amd.quality_tables(50)[0]
plane = torch.randint(0, 256, (8192, 8192), dtype=torch.uint8, device=dev)
for name, fn in (("encode_plane", lambda: jpeg_amd.encode_plane(plane, q)),):
    c = fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c = fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "us per 8192^2 plane:", e0.elapsed_time(e1) / 20 * 1e3, "-> x1.5 for a 4:2:0 image")
coef = jpeg_amd.encode_plane(plane, q)
out = jpeg_amd.decode_plane(coef, 8192, 8192, q); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): out = jpeg_amd.decode_plane(coef, 8192, 8192, q)
e1.record(); torch.cuda.synchronize()
print("decode_plane us per plane:", e0.elapsed_time(e1) / 20 * 1e3)
