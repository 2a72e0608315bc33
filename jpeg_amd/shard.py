"""MCU-row strip sharding of one image across ranks.

Every 16x16 MCU (4 Y + Cb + Cr blocks) is independent up to and including the
per-block run-length pairs (encode_decode.cpp:350-373: runs never cross
blocks, no DC prediction), so a rank encodes rows [r0, r1) of the image on its
own and the concatenation of the ranks' coefficient planes in rank order IS the
whole image's planes in block raster order (encode_decode.cpp:643-645).  The
only exchange is the gather of the encoded strips to the writer rank
(compress_image_sharded below); there is no data-path collective.

The functions here are host logic over torch tensors and torch.distributed
(backend nccl = RCCL on the GPUs, gloo in the CPU tests); the compute of a
strip is jpeg_amd.encode_image / decode_image on that rank's GPU.
"""
import torch

MCU = 16


def strip_rows(rows, world, rank):
    """Pixel rows [r0, r1) owned by `rank`: contiguous MCU rows, sizes differ by at
    most one MCU row; a rank may own nothing when world > rows/16."""
    if rows % MCU:
        raise ValueError("rows must be a multiple of 16 (ensureMultipleOf16, encode_decode.cpp:29-37)")
    if not 0 <= rank < world:
        raise ValueError("rank outside the world")
    n = rows // MCU
    return MCU * (rank * n // world), MCU * ((rank + 1) * n // world)


def strip_block_counts(rows, cols, world, rank):
    """(Y blocks, chroma blocks per plane) of the rank's strip."""
    r0, r1 = strip_rows(rows, world, rank)
    h = r1 - r0
    return (h // 8) * (cols // 8), (h // 16) * (cols // 16)


def split_planes(coef_y, coef_cb, coef_cr, rows, cols, world, rank):
    """The rank's strip of whole-image coefficient planes (views, no copy):
    what the root scatters before the inverse transform."""
    offs_y = offs_c = 0
    for g in range(rank):
        ny, nc = strip_block_counts(rows, cols, world, g)
        offs_y += ny
        offs_c += nc
    ny, nc = strip_block_counts(rows, cols, world, rank)
    return coef_y[offs_y:offs_y + ny], coef_cb[offs_c:offs_c + nc], coef_cr[offs_c:offs_c + nc]


def gather_planes(strip_y, strip_cb, strip_cr, rows, cols, group=None):
    """All ranks' strips -> whole-image planes on every rank, in rank order.
    NOT part of the compression path: dense planes are 10-40 x the encoded strips, so
    compress_image_sharded below gathers (value, run) streams to the writer instead.  Kept as the
    check that strips concatenate to the whole image's planes (tests/test_shard.py) and for
    callers that want the planes themselves on every rank.

    Strips differ in size by at most one MCU row, so one all_gather on buffers
    padded to the largest strip moves everything (over xGMI each rank's strip is
    one direct peer write per destination); the counts are known from the
    partition and need no exchange."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = [strip_block_counts(rows, cols, world, g) for g in range(world)]
    max_y = max(c[0] for c in counts)
    max_c = max(c[1] for c in counts)
    rank = dist.get_rank(group)
    if (strip_y.shape[0], strip_cb.shape[0], strip_cr.shape[0]) != (counts[rank][0], counts[rank][1], counts[rank][1]):
        raise ValueError("strip planes do not match this rank's share of the partition")
    # one buffer per rank: [Y | Cb | Cr], padded
    mine = torch.zeros((max_y + 2 * max_c, 64), dtype=strip_y.dtype, device=strip_y.device)
    mine[:strip_y.shape[0]] = strip_y
    mine[max_y:max_y + strip_cb.shape[0]] = strip_cb
    mine[max_y + max_c:max_y + max_c + strip_cr.shape[0]] = strip_cr
    # moved as raw bytes: gloo has no int16 collectives, and the payload is opaque anyway
    raw = [torch.empty_like(mine).view(torch.uint8) for _ in range(world)]
    dist.all_gather(raw, mine.view(torch.uint8), group=group)
    bufs = [r.view(mine.dtype) for r in raw]
    ys = [b[:c[0]] for b, c in zip(bufs, counts)]
    cbs = [b[max_y:max_y + c[1]] for b, c in zip(bufs, counts)]
    crs = [b[max_y + max_c:max_y + max_c + c[1]] for b, c in zip(bufs, counts)]
    return torch.cat(ys), torch.cat(cbs), torch.cat(crs)


def encode_image_sharded(bgr_strip, rows, cols, q_luma, q_chroma, group=None):
    """Each rank passes ITS strip of the image (rows strip_rows(...) of it, on its
    GPU); returns the whole image's coefficient planes on every rank (gather_planes: see there --
    to write a .bin use compress_image_sharded)."""
    import torch.distributed as dist
    from . import api
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    r0, r1 = strip_rows(rows, world, rank)
    if tuple(bgr_strip.shape) != (r1 - r0, cols, 3):
        raise ValueError("bgr_strip is not this rank's strip")
    if r1 > r0:
        sy, scb, scr = api.encode_image(bgr_strip, q_luma, q_chroma)
    else:
        sy, scb, scr = (torch.empty((0, 64), dtype=torch.int16, device=bgr_strip.device) for _ in range(3))
    return gather_planes(sy, scb, scr, rows, cols, group)


def decode_image_sharded(coef_y, coef_cb, coef_cr, rows, cols, q_luma, q_chroma, group=None):
    """Every rank holds the whole planes (after the root's entropy decode and a
    broadcast); each decodes its own strip and returns it (rows strip_rows(...))."""
    import torch.distributed as dist
    from . import api
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    r0, r1 = strip_rows(rows, world, rank)
    sy, scb, scr = split_planes(coef_y, coef_cb, coef_cr, rows, cols, world, rank)
    if r1 == r0:
        return torch.empty((0, cols, 3), dtype=torch.uint8, device=coef_y.device)
    return api.decode_image(sy.contiguous(), scb.contiguous(), scr.contiguous(), r1 - r0, cols, q_luma, q_chroma)


# ---------------------------------------------------------------------------------------------
# One .bin from N ranks: what crosses the links is the ENCODED strips (north_star
# "RCCL ... only to gather encoded strips").  Per rank: fused encode + run-length stage of its
# strip -> three (value, run) pair streams.  Sizes go round by all_gather (3 integers per rank),
# the streams go to the writer rank only, point to point (one hop per peer over xGMI; gloo in
# the CPU tests).  Concatenated in rank order they ARE the whole image's streams in block raster
# order (encode_decode.cpp:643-645), so the writer's entropy stage sees the symbols in the order
# the reference's build_frequency_dict does (:375-383) and writes the same bytes as one rank would.


def gather_streams_to_writer(streams, group=None, writer=0):
    """streams: this rank's three pair streams, int32 tensors [n_k, 2] (possibly empty), all on
    one device.  Returns, on the writer, the three whole-image streams (rank order); None elsewhere."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = streams[0].device
    flat = [s.reshape(-1, 2).to(torch.int32) for s in streams]
    mine = torch.tensor([int(f.shape[0]) for f in flat], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    sizes = [[int(v) for v in t.tolist()] for t in sizes]
    if rank != writer:
        if sum(sizes[rank]):
            dist.send(torch.cat(flat).contiguous(), dst=dist.get_global_rank(group, writer) if group else writer, group=group)
        return None
    parts = [[None] * world for _ in range(3)]
    for r in range(world):
        if r == writer:
            got = flat
        else:
            n = sum(sizes[r])
            buf = torch.empty((n, 2), dtype=torch.int32, device=dev)
            if n:
                dist.recv(buf, src=dist.get_global_rank(group, r) if group else r, group=group)
            a, b = sizes[r][0], sizes[r][0] + sizes[r][1]
            got = [buf[:a], buf[a:b], buf[b:]]
        for k in range(3):
            parts[k][r] = got[k]
    return [torch.cat(p) for p in parts]


def compress_image_sharded(bgr_strip, rows, cols, q_luma, q_chroma, group=None, writer=0, encode_strip=None):
    """Every rank passes ITS strip of the image (rows strip_rows(...)); the writer returns the
    bytes of the whole image's .bin (the reference's container, bit-compatible), the others None.
    encode_strip(strip) -> three pair streams; default: jpegamd_encode_image_rle_dev on the
    strip's GPU (the CPU tests pass a stand-in, they test the exchange and the writer)."""
    import torch.distributed as dist
    from . import api
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    r0, r1 = strip_rows(rows, world, rank)
    if tuple(bgr_strip.shape) != (r1 - r0, cols, 3):
        raise ValueError("bgr_strip is not this rank's strip")
    if encode_strip is None:
        encode_strip = lambda strip: [p for p, _ in api.encode_image_rle(strip, q_luma, q_chroma)[1]]
    if r1 > r0:
        streams = [s.reshape(-1, 2) for s in encode_strip(bgr_strip)]
    else:
        streams = [torch.empty((0, 2), dtype=torch.int32, device=bgr_strip.device) for _ in range(3)]
    whole = gather_streams_to_writer(streams, group, writer)
    if whole is None:
        return None
    if whole[0].is_cuda:
        return api.bin_encode_dev(*whole, rows, cols)
    return api.bin_encode(*(w.numpy() for w in whole), rows, cols)


def scatter_planes_from_writer(planes, rows, cols, group=None, writer=0, device=None):
    """The inverse exchange for decoding: the writer holds the whole image's coefficient planes
    (after its entropy and run-length decode); every rank receives its strip of them."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank == writer:
        mine = None
        for r in range(world):
            strip = [t.contiguous() for t in split_planes(*planes, rows, cols, world, r)]
            if r == writer:
                mine = strip
            elif strip[0].numel():
                dist.send(torch.cat([t.reshape(-1) for t in strip]).view(torch.uint8),
                          dst=dist.get_global_rank(group, r) if group else r, group=group)
        return mine
    ny, nc = strip_block_counts(rows, cols, world, rank)
    buf = torch.empty((ny + 2 * nc) * 64, dtype=torch.int16, device=device)
    if buf.numel():
        dist.recv(buf.view(torch.uint8), src=dist.get_global_rank(group, writer) if group else writer, group=group)
    return [buf[:ny * 64].view(ny, 64), buf[ny * 64:(ny + nc) * 64].view(nc, 64), buf[(ny + nc) * 64:].view(nc, 64)]
