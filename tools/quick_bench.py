"""Developer helper: kernel timings (HIP events) + a fast parity spot check."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import jpeg_amd
from jpeg_amd.synth import synth_image

def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    jpeg_amd.init(0)
    dev = torch.device("cuda:0")
    qy, qc = jpeg_amd.quality_tables(50)
    # parity spot check against the CPU oracle
    import oracle_lib
    orc = oracle_lib.Oracle()
    small = synth_image(64, 1040, 3, "photo", device=dev)
    coefs = jpeg_amd.encode_image(small, qy, qc)
    want = orc.encode_image(small.cpu().numpy(), qy, qc)
    bad = sum(int((c.cpu().numpy() != w).sum()) for c, w in zip(coefs, want))
    rec = jpeg_amd.decode_image(*coefs, 64, 1040, qy, qc).cpu().numpy()
    wpx = orc.decode_image(*[c.cpu().numpy() for c in coefs], 64, 1040, qy, qc)
    print("spot check: coef mismatches", bad, "pixel mismatches", int((rec != wpx).sum()), "max", int(np.abs(rec.astype(int) - wpx).max()))
    for kind in ("photo", "noise"):
        bgr = synth_image(size, size, 0x8192, kind, device=dev)
        coefs = jpeg_amd.encode_image(bgr, qy, qc)
        rec = torch.empty_like(bgr)
        jpeg_amd.time_image(bgr, coefs, rec, qy, qc, 5)
        res = [jpeg_amd.time_image(bgr, coefs, rec, qy, qc, iters) for _ in range(3)]
        e = min(r[0] for r in res); d = min(r[1] for r in res)
        npix = size * size
        print(json.dumps({"kind": kind, "enc_us": e * 1e3, "dec_us": d * 1e3,
                          "enc_GBs": 6 * npix / e / 1e6, "dec_GBs": 6 * npix / d / 1e6,
                          "frac_of_8TBs": 12 * npix / (e + d) / 1e6 / 8000}))

main()
