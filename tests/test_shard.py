"""MCU-row strip sharding (jpeg_amd/shard.py).

CPU: the partition's arithmetic, and a world_size-2 gloo run in which each rank
encodes its strip (the oracle stands in for the GPU: this tests the host logic
and the exchange, not the kernels) and the gathered planes must equal the
whole-image encode bit for bit.  GPU: the same with virtual ranks on one card,
through the real kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from jpeg_amd import shard
from jpeg_amd.synth import synth_image

HERE = os.path.dirname(os.path.abspath(__file__))


def test_strip_partition_covers_image_once():
    for rows in (16, 32, 48, 1024, 2160 // 16 * 16, 8192):
        for world in (1, 2, 3, 4, 7, 8, 600):
            edges = [shard.strip_rows(rows, world, g) for g in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == rows
            for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [(b - a) // 16 for a, b in edges]
            assert max(sizes) - min(sizes) <= 1 and all(a % 16 == 0 for a, _ in edges)
    with pytest.raises(ValueError):
        shard.strip_rows(100, 2, 0)
    with pytest.raises(ValueError):
        shard.strip_rows(64, 2, 2)


def test_split_planes_is_block_raster_order():
    rows, cols, world = 80, 64, 3
    ny, nc = (rows // 8) * (cols // 8), (rows // 16) * (cols // 16)
    cy = torch.arange(ny).reshape(-1, 1).repeat(1, 64)
    cb = torch.arange(nc).reshape(-1, 1).repeat(1, 64)
    got_y, got_c = [], []
    for g in range(world):
        sy, scb, scr = shard.split_planes(cy, cb, cb, rows, cols, world, g)
        r0, r1 = shard.strip_rows(rows, world, g)
        assert sy.shape[0] == (r1 - r0) // 8 * (cols // 8) and scb.shape[0] == (r1 - r0) // 16 * (cols // 16)
        got_y.append(sy)
        got_c.append(scb)
    assert torch.equal(torch.cat(got_y), cy) and torch.equal(torch.cat(got_c), cb)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_worker(rank, world, port, rows, cols, result_path):
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import oracle_lib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_lib.Oracle()
        qy, qc = orc.quality_tables(50)
        img = synth_image(rows, cols, 7, "photo").numpy()
        r0, r1 = shard.strip_rows(rows, world, rank)
        if r1 > r0:
            sy, scb, scr = (torch.from_numpy(a) for a in orc.encode_image(img[r0:r1], qy, qc, threads=2))
        else:
            sy = scb = scr = torch.empty((0, 64), dtype=torch.int16)
        gy, gcb, gcr = shard.gather_planes(sy, scb, scr, rows, cols)
        # decode side: every rank slices its strip back out of the whole planes
        by, bcb, bcr = shard.split_planes(gy, gcb, gcr, rows, cols, world, rank)
        ok_back = torch.equal(by, sy) and torch.equal(bcb, scb) and torch.equal(bcr, scr)
        if rank == 0:
            wy, wcb, wcr = orc.encode_image(img, qy, qc, threads=2)
            ok = (np.array_equal(gy.numpy(), wy) and np.array_equal(gcb.numpy(), wcb)
                  and np.array_equal(gcr.numpy(), wcr))
            with open(result_path, "w") as f:
                f.write("ok" if ok and ok_back else "mismatch")
        flag = torch.tensor([1 if ok_back else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        assert int(flag) == 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows", [96, 112])  # 6 and 7 MCU rows: even and uneven strips
def test_gloo_world2_gather_equals_whole_image(tmp_path, rows):
    import torch.multiprocessing as mp
    world, cols = 2, 160
    result = tmp_path / "result.txt"
    mp.spawn(_gloo_worker, args=(world, _free_port(), rows, cols, str(result)), nprocs=world, join=True)
    assert result.read_text() == "ok"


def test_gloo_world3_with_an_empty_rank(tmp_path):
    import torch.multiprocessing as mp
    result = tmp_path / "result.txt"
    mp.spawn(_gloo_worker, args=(3, _free_port(), 32, 64, str(result)), nprocs=3, join=True)
    assert result.read_text() == "ok"


def _bin_worker(rank, world, port, case, result_path):
    """every rank encodes its strip (oracle as the stand-in for the GPU), the encoded strips go to
    the writer, the writer's .bin must be the single-rank file byte for byte"""
    sys.path.insert(0, HERE)
    import hashlib
    import torch.distributed as dist
    import oracle_lib
    import jpeg_amd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_lib.Oracle()
        qy, qc = orc.quality_tables(50)
        if case == "lena":
            from PIL import Image
            rgb = np.array(Image.open(os.path.join(HERE, "golden", "lena_color_1024.png")).convert("RGB"))
            img = np.ascontiguousarray(rgb[:, :, ::-1])
        else:
            rows, cols = case
            img = synth_image(rows, cols, 9, "photo").numpy()
        rows, cols = img.shape[:2]
        r0, r1 = shard.strip_rows(rows, world, rank)

        def encode_strip(strip):
            coefs = orc.encode_image(strip.numpy(), qy, qc, threads=2)
            return [torch.from_numpy(orc.rle_stream(c).reshape(-1, 2).copy()) for c in coefs]

        writer = world - 1 if case != "lena" else 0    # the writer need not be rank 0
        data = shard.compress_image_sharded(torch.from_numpy(img[r0:r1].copy()), rows, cols, qy, qc,
                                            writer=writer, encode_strip=encode_strip)
        assert (data is None) == (rank != writer)
        if rank == writer:
            whole = orc.write_bin(*(orc.rle_stream(c) for c in orc.encode_image(img, qy, qc, threads=2)), rows, cols)
            ok = data == whole
            if case == "lena":
                gold = open(os.path.join(HERE, "golden", "compressed_image_cpu_50.bin"), "rb").read()
                ok = ok and data == gold and hashlib.sha256(data).hexdigest().startswith("7870d3fec500635c")
            # and back: the writer decodes the container, the strips of the planes go out again
            dims, syms = jpeg_amd.bin_decode(data)
            nbs = ((rows // 8) * (cols // 8), (rows // 16) * (cols // 16), (rows // 16) * (cols // 16))
            planes = [torch.from_numpy(orc.decode_rle(s, nb)[0]) for s, nb in zip(syms, nbs)]
        else:
            planes = None
        strip = shard.scatter_planes_from_writer(planes, rows, cols, writer=writer)
        want = orc.encode_image(img[r0:r1], qy, qc, threads=2) if r1 > r0 else [np.zeros((0, 64), np.int16)] * 3
        back = all(np.array_equal(s.numpy(), w) for s, w in zip(strip, want))
        flag = torch.tensor([1 if back else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == writer:
            with open(result_path, "w") as f:
                f.write("ok" if ok and int(flag) == 1 else f"mismatch bin={ok} planes={int(flag)}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, (96, 160)), (3, (32, 64)), (2, "lena"), (3, "lena")])
def test_gloo_encoded_strips_to_the_writer_give_the_single_rank_bin(tmp_path, world, case):
    """N ranks, ONE .bin, byte-identical to the single-rank file (and, for the
    reference's Lena, to its committed golden); world 3 on a 32-row image has an empty rank."""
    import torch.multiprocessing as mp
    result = tmp_path / "result.txt"
    mp.spawn(_bin_worker, args=(world, _free_port(), case, str(result)), nprocs=world, join=True)
    assert result.read_text() == "ok"


@pytest.mark.gpu
def test_virtual_ranks_on_one_gpu_equal_whole_image(gpu):
    """G strips encoded independently by the kernels concatenate to the whole-image
    planes bit for bit, and strip decodes tile the whole-image decode."""
    import jpeg_amd
    rows, cols = 1072, 1040  # 67 MCU rows, partial 512-px tiles
    qy, qc = jpeg_amd.quality_tables(50)
    img = synth_image(rows, cols, 11, "photo", device=gpu)
    whole = jpeg_amd.encode_image(img, qy, qc)
    rec = jpeg_amd.decode_image(*whole, rows, cols, qy, qc)
    for world in (2, 3, 8):
        parts = []
        for g in range(world):
            r0, r1 = shard.strip_rows(rows, world, g)
            parts.append(jpeg_amd.encode_image(img[r0:r1].contiguous(), qy, qc))
            s = shard.split_planes(*whole, rows, cols, world, g)
            strip = jpeg_amd.decode_image(*(t.contiguous() for t in s), r1 - r0, cols, qy, qc)
            assert torch.equal(strip, rec[r0:r1])
        for k in range(3):
            assert torch.equal(torch.cat([p[k] for p in parts]), whole[k])


@pytest.mark.gpu
def test_virtual_ranks_encoded_strips_give_the_single_gpu_bin(gpu):
    """the strips' (value, run) streams from the real kernels (jpegamd_encode_image_rle_dev per
    strip), concatenated in rank order and entropy-coded on the device, are the .bin of the
    whole image through jpegamd_compress_image"""
    import jpeg_amd
    rows, cols = 528, 1040
    qy, qc = jpeg_amd.quality_tables(50)
    img = synth_image(rows, cols, 13, "photo", device=gpu)
    want = jpeg_amd.compress_image(img.cpu().numpy(), qy, qc)
    for world in (1, 2, 5, 40):  # 33 MCU rows: 40 ranks leave some without a strip
        streams = [[], [], []]
        for g in range(world):
            r0, r1 = shard.strip_rows(rows, world, g)
            if r1 > r0:
                for k, (pairs, _) in enumerate(jpeg_amd.encode_image_rle(img[r0:r1].contiguous(), qy, qc)[1]):
                    streams[k].append(pairs)
        data = jpeg_amd.bin_encode_dev(*(torch.cat(s) for s in streams), rows, cols)
        assert data == want, world


@pytest.mark.gpu
def test_compress_image_sharded_over_rccl_world_of_one(gpu, lena_bgr):
    """compress_image_sharded had never run on the `nccl` backend.  A world of ONE rank
    on the one GPU executes exactly the calls the 8-GPU run makes -- init over RCCL, all_gather of
    device tensors for the stream sizes, the writer's concatenation and device entropy stage (only the
    send/recv pairs need a second rank; gloo world-2/3 above cover those).  The file must be the
    single-call one and, for Lena, the reference's committed golden .bin."""
    import hashlib
    import torch.distributed as dist
    import jpeg_amd
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", world_size=1, rank=0,
                            device_id=gpu)
    try:
        qy, qc = jpeg_amd.quality_tables(50)
        data = shard.compress_image_sharded(torch.from_numpy(lena_bgr).to(gpu), 1024, 1024, qy, qc)
        assert hashlib.sha256(data).hexdigest() == "7870d3fec500635c7ef9dc5a98f985f31dbb2c91b0beb8d0c6ba9f9dbf82b2bf"
        rows, cols = 528, 1040
        img = synth_image(rows, cols, 13, "photo", device=gpu)
        assert shard.compress_image_sharded(img, rows, cols, qy, qc) == jpeg_amd.compress_image(img.cpu().numpy(), qy, qc)
        # the dense-plane exchange of the decode direction on the same backend
        whole = jpeg_amd.encode_image(img, qy, qc)
        mine = shard.scatter_planes_from_writer(list(whole), rows, cols, device=gpu)
        assert all(torch.equal(a, b) for a, b in zip(mine, whole))
        t = torch.ones(3, device=gpu)
        dist.all_reduce(t)  # and one plain RCCL collective on device memory
        assert t.tolist() == [1.0, 1.0, 1.0]
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_bench_strips_mode_gives_the_same_bin_for_one_and_two_ranks(gpu):
    """bench.py --mode strips: ONE image, MCU-row strips over the ranks, encoded strips gathered to the
    writer, one .bin.  N = 1 runs over RCCL (a world of one); N = 2 on this one-GPU box is the rehearsal
    (ranks share the GPU, exchange over gloo).  Same sha256, and the same as compress_image of the image."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    import jpeg_amd
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    lines = []
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "strips", "--gpus", str(n), "--size", "1024",
                            "--steps", "2", "--warmup", "1", "--repeats", "1"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert lines[0]["n_gpus"] == 1 and lines[1]["n_gpus"] == 2 and "REHEARSAL" in lines[1]["config"]["parallelism"]
    assert lines[0]["bin_sha256"] == lines[1]["bin_sha256"]
    qy, qc = jpeg_amd.quality_tables(50)
    want = jpeg_amd.compress_image(synth_image(1024, 1024, 0x8192, "photo").numpy(), qy, qc)
    assert lines[0]["bin_sha256"] == hashlib.sha256(want).hexdigest() and lines[0]["bin_bytes"] == len(want)
