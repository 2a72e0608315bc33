"""Developer helper: the bench's step sequence (encode, decode, encode, ...) software-pipelined over two streams with
double-buffered coefficient planes -- encode of step k+1 may overlap decode of step k -- against the strict sequence on
one stream.  Same work, same dependencies (a decode waits for ITS encode), wall clock per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, jpeg_amd
from jpeg_amd.synth import synth_image
size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = 50
jpeg_amd.init(0)
dev = torch.device("cuda:0")
qy, qc = jpeg_amd.quality_tables(50)
bgr = synth_image(size, size, 0x8192, "photo", device=dev)
coefs = [jpeg_amd.encode_image(bgr, qy, qc) for _ in range(2)]
rec = [jpeg_amd.decode_image(*coefs[0], size, size, qy, qc) for _ in range(2)]
ref = rec[0].clone()
torch.cuda.synchronize()


def serial(n):
    for _ in range(n):
        jpeg_amd.encode_image(bgr, qy, qc, out=coefs[0])
        jpeg_amd.decode_image(*coefs[0], size, size, qy, qc, out=rec[0])


s_enc, s_dec = torch.cuda.Stream(), torch.cuda.Stream()


def pipelined(n):
    enc_done = [torch.cuda.Event() for _ in range(n)]
    dec_done = [torch.cuda.Event() for _ in range(n)]
    for k in range(n):
        b = k & 1
        if k >= 2:
            s_enc.wait_event(dec_done[k - 2])  # the decode that last read this coefficient buffer
        jpeg_amd.encode_image(bgr, qy, qc, out=coefs[b], stream=s_enc.cuda_stream)
        enc_done[k].record(s_enc)
        s_dec.wait_event(enc_done[k])
        jpeg_amd.decode_image(*coefs[b], size, size, qy, qc, out=rec[b], stream=s_dec.cuda_stream)
        dec_done[k].record(s_dec)


for name, fn in (("serial", serial), ("pipelined", pipelined), ("serial", serial), ("pipelined", pipelined)):
    fn(5)
    torch.cuda.synchronize()
    res = []
    for _ in range(5):
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e6)
    ok = torch.equal(rec[0], ref) and torch.equal(rec[1], ref)
    print(f"{name:10s} {sorted(res)[2]:7.1f} us per step (five regions of {steps}: {' '.join('%.1f' % r for r in res)})  "
          f"{size * size / sorted(res)[2]:.0f} Mpixel/s  outputs equal: {ok}")
