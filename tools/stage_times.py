"""Developer helper: wall time of every stage of the image -> .bin -> image path at one size
(device stages with torch.cuda events around them, host stages with perf_counter)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, jpeg_amd
from jpeg_amd.synth import synth_image

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
img = synth_image(size, size, 0x8192, "photo", device=dev)


def dev_time(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


res = {}
res["encode_image_ms"], coefs = dev_time(lambda: jpeg_amd.encode_image(img, qy, qc))
res["rle_encode_3planes_ms"], pr = dev_time(lambda: [jpeg_amd.rle_encode_plane(c) for c in coefs])
res["pairs_per_block"] = [round(p[0].shape[0] / c.shape[0], 2) for p, c in zip(pr, coefs)]
t0 = time.perf_counter(); host = [p[0].cpu().numpy().reshape(-1) for p in pr]; res["pairs_d2h_ms"] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); data = jpeg_amd.bin_encode(*host, size, size); res["bin_encode_host_ms"] = (time.perf_counter() - t0) * 1e3
res["bin_bytes"] = len(data)
t0 = time.perf_counter(); dims, syms = jpeg_amd.bin_decode(data); res["bin_decode_host_ms"] = (time.perf_counter() - t0) * 1e3
dp = [torch.from_numpy(s[:s.size // 2 * 2].reshape(-1, 2).copy()).to(dev) for s in syms]
res["rle_decode_3planes_ms"], planes = dev_time(lambda: [jpeg_amd.rle_decode_plane(p, c.shape[0]) for p, c in zip(dp, coefs)])
assert all(torch.equal(a, b) for a, b in zip(planes, coefs))
res["decode_image_ms"], rec = dev_time(lambda: jpeg_amd.decode_image(*planes, size, size, qy, qc))
himg = img.cpu().numpy()
t0 = time.perf_counter(); d2 = jpeg_amd.compress_image(himg, qy, qc); res["compress_image_host_to_bin_ms"] = (time.perf_counter() - t0) * 1e3
assert d2 == data
t0 = time.perf_counter(); r2 = jpeg_amd.decompress_image(data, qy, qc); res["decompress_image_bin_to_host_ms"] = (time.perf_counter() - t0) * 1e3
assert np.array_equal(r2, rec.cpu().numpy())
print(json.dumps(res))
