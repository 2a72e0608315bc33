"""Generates tests/golden/ref_block_vectors.npz and wikipedia_block.json from the
REFERENCE'S OWN code (oracle/_ref = jpeg_concept/jpeg_function.cpp compiled where
it lies).  Run in the build container (needs /root/reference); the outputs are
data -- inputs and the reference's outputs -- and travel to the GPU box.

    make -C oracle && python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402


def main():
    ref = oracle_lib.Reference()
    rng = np.random.default_rng(20250218)
    px, qf, dct, quant, zz, rle_flat, rle_len, restored = [], [], [], [], [], [], [], []
    kinds = ["uniform", "smooth", "flat", "edge", "extreme"]
    for n in range(400):
        kind = kinds[n % len(kinds)]
        if kind == "uniform":
            p = rng.integers(0, 256, 64)
        elif kind == "smooth":
            yy, xx = np.mgrid[0:8, 0:8]
            p = 128 + rng.uniform(-100, 100) * np.cos((2 * xx + 1) * rng.integers(0, 3) * np.pi / 16) \
                * np.cos((2 * yy + 1) * rng.integers(0, 3) * np.pi / 16) + rng.normal(0, 3, (8, 8))
            p = np.clip(np.rint(p), 0, 255).reshape(64)
        elif kind == "flat":
            p = np.full(64, rng.integers(0, 256))        # DC-only: the tie-prone case
        elif kind == "edge":
            p = np.where(np.arange(64) % 8 < rng.integers(1, 8), rng.integers(0, 256), rng.integers(0, 256))
        else:
            p = rng.choice([0, 255], 64)
        p = p.astype(np.int32)
        q = int(rng.choice([1, 5, 10, 25, 50, 75, 90, 95, 100]))
        table = ref.luma_table(q)
        d = ref.recenter_dct(p)
        c = ref.quantize(d, table)
        z = ref.zigzag(c)
        r = ref.rle(z)
        back = ref.decode_block(z, table)
        px.append(p), qf.append(q), dct.append(d), quant.append(c), zz.append(z)
        rle_flat.extend(r.tolist()), rle_len.append(len(r)), restored.append(back)
    sym = np.array(rle_flat, np.int32)
    bits = ref.huffman_encode(sym)
    order = ref.freq_dict_order(sym)
    np.savez_compressed(
        os.path.join(HERE, "ref_block_vectors.npz"),
        pixels=np.array(px, np.int32), quality=np.array(qf, np.int32), dct=np.array(dct, np.float32),
        quantised=np.array(quant, np.float32), zigzag=np.array(zz, np.int32),
        rle=sym, rle_len=np.array(rle_len, np.int32), restored=np.array(restored, np.int32),
        huffman_bits=np.frombuffer(bits.encode(), np.uint8), freq_dict_order=order)
    print("blocks", len(px), "symbols", sym.size, "bits", len(bits), "dict", len(order))


if __name__ == "__main__":
    main()
