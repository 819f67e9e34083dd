"""Ray-batch sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in CPU tests and in the single-device rehearsal).

The reference is single-GPU (``examples/train_finetune.py:218`` pins ``cuda:0``); this is
new.  Rays are independent, every rank holds a replica of mesh/BVH + tables + weights, and ONE frame is cut into
contiguous ROW BANDS, one per rank:

* a band ``[y0, y1)`` of a pinhole frame is itself the full pixel grid of a pinhole camera -- the same camera with
  the principal point moved up by ``y0`` rows (``band_camera``) -- so the camera-coherent intersector, the 8x8-tile
  processing order and the streamed field path (the whole of the single-GPU speed) apply to a band unchanged, and a
  band's rays are a contiguous slice of the frame's ray arrays;
* every per-ray result depends on that ray alone (hits, samples, field values, the compositing sum in rank order),
  so the bands' pixels are bit-identical to the 1-rank frame whatever the cuts are (tested on the GPU);
* the cuts are multiples of 8 rows (tile rows stay tile rows) and are balanced by a per-row cost profile taken from
  the per-row SAMPLE COUNTS of an earlier frame, which the bands report in an extra padded row of the frame's own
  collective (``COST_RAY`` per pixel + ``COST_SAMPLE`` per quadrature point; gathered alpha only stands in when a
  renderer reports no counts).  The profile is applied with a fixed lag, so every rank cuts every frame identically
  without a further collective;
* the finished bands (rgb3 + alpha1 + depth1 = 20 B/ray) are exchanged with ONE collective per frame:
  ``all_to_all_single`` with split sizes (every band arrives at its own length, in place: 12.8 MB on the wire at
  800x800, latency- rather than bandwidth-bound on xGMI); the round-2 ``all_gather_into_tensor`` of bands padded to the
  tallest one remains selectable (``GATHER_MODE``).

The 8x8-tile round-robin sharding of round 1 (``shard_tiles`` ...) is kept for arbitrary (non-pinhole) ray sets.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

TILE = 8
BAND_ALIGN = 8
#: relative cost of a row: COST_RAY per pixel + COST_SAMPLE per quadrature point of the row (the bands report their
#: per-row sample counts along with their pixels).  From the single-GPU stage times: 0.39 ns of pack + field +
#: compositing per point, ~0.15 ns of set-up per ray.  COST_OBJECT per pixel that hit the object (alpha > 0) stands in
#: for the points (~10 per object ray) when a renderer reports no counts.
COST_RAY, COST_SAMPLE, COST_OBJECT = 0.15, 0.39, 3.9


# ----------------------------------------------------------------------------------------------------------
# row bands (pinhole frames)

def band_cuts(height: int, world_size: int, row_cost: Optional[Sequence[float]] = None,
              align: int = BAND_ALIGN) -> List[int]:
    """``world_size + 1`` ascending row boundaries (multiples of ``align`` except the last = ``height``): rank r
    renders rows ``[cuts[r], cuts[r+1])``.  ``row_cost`` [height]: relative cost per row (None = uniform); the cuts
    equalise the cost per rank.  Every rank gets at least one block of rows while there are blocks left."""
    n_blocks = (height + align - 1) // align
    if row_cost is None:
        cost = np.ones(n_blocks, dtype=np.float64)
    else:
        rc = np.asarray(row_cost, dtype=np.float64).reshape(-1)
        if rc.shape[0] != height:
            raise ValueError("row_cost must have one entry per row")
        pad = n_blocks * align - height
        cost = np.pad(np.maximum(rc, 0.0), (0, pad)).reshape(n_blocks, align).sum(axis=1)
    cost = cost + 1e-9 * max(float(cost.sum()), 1.0) / n_blocks + 1e-30      # strictly increasing prefix
    prefix = np.concatenate([[0.0], np.cumsum(cost)])
    cuts_b = [0]
    for r in range(1, world_size):
        target = prefix[-1] * r / world_size
        b = int(np.searchsorted(prefix, target, side="left"))
        if b > 0 and target - prefix[b - 1] < prefix[b] - target:
            b -= 1                                                          # the nearer block boundary
        if n_blocks >= world_size:          # at least one block for every rank before and after this cut
            b = min(max(b, cuts_b[-1] + 1), n_blocks - (world_size - r))
        else:                               # fewer blocks than ranks: the last ranks get empty bands
            b = min(max(b, cuts_b[-1]), n_blocks)
        cuts_b.append(int(b))
    cuts_b.append(n_blocks)
    return [min(b * align, height) for b in cuts_b]


def band_camera(c2w, focal: float, width: int, height: int, y0: int, y1: int):
    """The pinhole camera whose full pixel grid is rows ``[y0, y1)`` of the ``width x height`` frame of
    ``mesh_utils.make_camera(c2w, focal, width, height)``: same pose and focal length, principal point moved up by
    ``y0`` rows.  The camera only bounds the intersector's search (of its values only the centre enters the hit
    arithmetic, and only once the pass has verified that it IS every ray's origin), so the band's hits are those of the
    frame's rays."""
    from .mesh_utils import make_camera
    cam = make_camera(c2w, focal, width, height)
    cam.cy = height / 2.0 - float(y0)
    cam.height = int(y1 - y0)
    # the band sees a part of the scene: have the camera-coherent pass cull the triangles in chunks before it projects
    # them (qf_raster_intersect cull_chunks), so that its set-up work shrinks with the band
    cam.cull = (y1 - y0) < height
    return cam


#: how the finished bands are exchanged.  "exact" (default): ``all_to_all_single`` with split sizes -- every rank sends
#: its band (+ its per-row numbers) to every rank and receives each band at its own length straight into place in the
#: frame: the bytes on the wire are the frame's own 20 B/ray, nothing is padded, nothing is concatenated afterwards.
#: "padded": one ``all_gather_into_tensor`` of the bands padded to the tallest (round 2; with cost-balanced cuts of an
#: 800-row frame over 8 ranks -- [224, 72, 56, 48, 48, 56, 72, 224] rows -- that moved 28.7 MB for a 12.8 MB frame).
#: "broadcast": one ``dist.broadcast`` per rank straight into place in the frame (the plainest primitive; the last
#: resort of ``selftest_collectives``, which picks the first of the three that delivers the right bytes on every rank).
GATHER_MODE = "exact"


def _meta_slots(rows: int, c: int) -> int:
    """dim-0 entries of a [*, c] array that hold ``rows`` per-row numbers"""
    return (rows + c - 1) // c


def gather_bands(local: torch.Tensor, cuts: Sequence[int], width: int, rank: int, world_size: int,
                 async_op: bool = False, meta: Optional[torch.Tensor] = None, mode: Optional[str] = None,
                 force_collective: bool = False):
    """local [rows_r * width, C] (this rank's band, row-major) -> the full frame [H * width, C] on every rank with ONE
    collective (see ``GATHER_MODE``).  ``async_op``: returns a zero-argument callable instead; the collective runs on
    the backend's own stream (RCCL) beside whatever is launched next, and calling the callable makes the current stream
    wait for it and hands out the frame.
    ``meta`` (every rank or none): float32 [rows_r], one number per row of the band (the band's per-row sample counts);
    it rides behind the band in the same collective and comes back as [H]: the result is then
    ``(frame, meta_of_every_row)``.
    ``force_collective``: issue the collective even in a one-rank group (``selftest_collectives``: a 1-rank RCCL group
    on a one-GPU box still runs the backend's real entry points)."""
    rows = [cuts[r + 1] - cuts[r] for r in range(world_size)]
    if local.shape[0] != rows[rank] * width:
        raise ValueError(f"rank {rank}: band has {local.shape[0]} rays, expected {rows[rank] * width}")
    if meta is not None and meta.numel() != rows[rank]:
        raise ValueError(f"rank {rank}: meta has {meta.numel()} entries, expected {rows[rank]}")
    if world_size == 1 and not force_collective:
        result = local if meta is None else (local, meta.to(torch.float32).reshape(-1))
        return (lambda: result) if async_op else result
    mode = GATHER_MODE if mode is None else mode
    c = local.shape[1]
    staged = local.is_cuda and dist.get_backend() == "gloo"     # rehearsal on one GPU: gloo moves host memory
    if mode == "exact":
        # chunk of rank r = its band followed by ceil(rows_r / C) entries of per-row numbers (only with meta)
        sizes = [rows[r] * width + (_meta_slots(rows[r], c) if meta is not None else 0) for r in range(world_size)]
        # the same chunk to every rank: the band (and its per-row numbers) written straight into the world_size copies
        # all_to_all_single wants -- two broadcast copies (13 MB at 800x800); the slots behind the numbers are never read
        send3 = torch.empty((world_size, sizes[rank], c), dtype=local.dtype, device=local.device)
        send3[:, :local.shape[0]] = local
        if meta is not None:
            send3[:, local.shape[0]:].reshape(world_size, -1)[:, :rows[rank]] = \
                meta.to(device=local.device, dtype=local.dtype).reshape(-1)
        send = send3.view(world_size * sizes[rank], c)
        if staged:
            send = send.cpu()
        buf = torch.empty((sum(sizes), c), dtype=send.dtype, device=send.device)
        work = dist.all_to_all_single(buf, send, output_split_sizes=sizes, input_split_sizes=[sizes[rank]] * world_size,
                                      async_op=async_op)

        def finish_exact():
            if work is not None:
                work.wait()                   # current stream waits for the collective; send / buf stay referenced here
            if meta is None:
                frame = buf                   # the bands lie back to back: this IS the frame
                return frame.to(local.device) if staged else frame
            starts = [sum(sizes[:r]) for r in range(world_size)]
            frame = torch.cat([buf[starts[r]:starts[r] + rows[r] * width] for r in range(world_size)], dim=0)
            m = torch.cat([buf[starts[r] + rows[r] * width:starts[r] + sizes[r]].reshape(-1)[:rows[r]]
                           for r in range(world_size)], dim=0)
            return (frame.to(local.device), m.to(local.device)) if staged else (frame, m)

        return finish_exact if async_op else finish_exact()
    if mode == "broadcast":
        # the plainest primitive: rank r broadcasts its chunk into its place of the frame buffer, one call per rank
        sizes = [rows[r] * width + (_meta_slots(rows[r], c) if meta is not None else 0) for r in range(world_size)]
        starts = [sum(sizes[:r]) for r in range(world_size)]
        host = torch.device("cpu") if staged else local.device
        buf = torch.empty((sum(sizes), c), dtype=local.dtype, device=host)
        mine = buf[starts[rank]:starts[rank] + sizes[rank]]
        mine[:local.shape[0]] = local
        if meta is not None:
            mine[local.shape[0]:].reshape(-1)[:rows[rank]] = meta.to(device=host, dtype=local.dtype).reshape(-1)
        works = [dist.broadcast(buf[starts[r]:starts[r] + sizes[r]], src=r, async_op=async_op)
                 for r in range(world_size) if sizes[r] > 0]

        def finish_broadcast():
            for w in works:
                if w is not None:
                    w.wait()
            if meta is None:
                return buf.to(local.device) if staged else buf      # the bands lie back to back: this IS the frame
            frame = torch.cat([buf[starts[r]:starts[r] + rows[r] * width] for r in range(world_size)], dim=0)
            m = torch.cat([buf[starts[r] + rows[r] * width:starts[r] + sizes[r]].reshape(-1)[:rows[r]]
                           for r in range(world_size)], dim=0)
            return (frame.to(local.device), m.to(local.device)) if staged else (frame, m)

        return finish_broadcast if async_op else finish_broadcast()
    if mode != "padded":
        raise ValueError(f"gather mode {mode!r}")
    band_cap = max(rows) * width
    meta_rays = 0 if meta is None else ((max(rows) + c - 1) // c + width - 1) // width * width      # whole padded rows
    cap = band_cap + meta_rays
    send = torch.zeros((cap, c), dtype=local.dtype, device=local.device)
    send[:local.shape[0]] = local
    if meta is not None:
        send[band_cap:].view(-1)[:rows[rank]] = meta.to(device=local.device, dtype=local.dtype).reshape(-1)
    if staged:
        send = send.cpu()
    send = send.contiguous()
    buf = torch.empty((world_size * cap, c), dtype=send.dtype, device=send.device)
    work = dist.all_gather_into_tensor(buf, send, async_op=async_op)

    def finish():
        if work is not None:
            work.wait()                       # current stream waits for the collective; send / buf stay referenced here
        b = buf.view(world_size, cap, c)
        frame = torch.cat([b[r, :rows[r] * width] for r in range(world_size)], dim=0)
        frame = frame.to(local.device) if staged else frame
        if meta is None:
            return frame
        m = torch.cat([b[r, band_cap:].reshape(-1)[:rows[r]] for r in range(world_size)], dim=0)
        return frame, (m.to(local.device) if staged else m)

    return finish if async_op else finish()


class ShardedFrameRenderer:
    """One frame per call, split into row bands over the ranks of the default process group and gathered on every
    rank: ``render(origins, viewdirs, c2w, focal, width, height) -> [H*W, 5]`` (rgb, alpha, depth).

    ``renderer`` is a ``render.FrameRenderer`` (this rank's replica of the scene).  ``origins`` / ``viewdirs`` are the
    frame's full row-major ray arrays (every rank generates or holds them; a band is a zero-copy slice).  The band
    cuts follow the cost profile of the frame rendered ``PROFILE_LAG`` calls earlier -- by then its 3 KB host copy
    has long arrived, and because the lag is fixed every rank uses the same profile for the same frame."""

    PROFILE_LAG = 2

    def __init__(self, renderer, rank: int = 0, world_size: int = 1, balance: bool = True):
        self.renderer, self.rank, self.world = renderer, int(rank), int(world_size)
        self.balance = bool(balance) and self.world > 1
        self._profiles = []          # (pinned host row counts, event), oldest first
        self.row_cost = None         # np [H] or None: the profile the next cuts are made from
        self.last_cuts = None
        self._ring = {}

    def cuts_for(self, height: int) -> List[int]:
        if self.row_cost is not None and self.row_cost.shape[0] != height:
            self.row_cost = None
        return band_cuts(height, self.world, self.row_cost)

    def render_band(self, origins, viewdirs, c2w, focal, width: int, height: int, y0: int, y1: int) -> torch.Tensor:
        """[(y1-y0)*W, 5] for rows [y0, y1) through the HIP path."""
        if y1 <= y0:
            return torch.empty((0, 5), dtype=torch.float32, device=origins.device)
        cam = band_camera(c2w, focal, width, height, y0, y1)
        o, d = origins[y0 * width:y1 * width], viewdirs[y0 * width:y1 * width]
        # no host wait inside a band when the renderer offers it (FrameRenderer.render_async): at 8 ranks a band is
        # ~0.3 ms of kernels, too short to hide a mid-frame round trip
        draw = getattr(self.renderer, "render_async", None)
        if callable(draw):
            local = draw(o, d, cam, packed=True)[0]            # [rows*W, 5] straight from the compositor
        else:
            rgb, alpha, depth, _ = self.renderer.render(o, d, camera=cam)
            local = torch.cat([rgb, alpha, depth], dim=1)
        # the band's quadrature points per row, for the next frames' cuts (a renderer without the hook reports none)
        hook = getattr(self.renderer, "row_samples", None)
        self._band_samples = hook() if (callable(hook) and self.balance) else None
        return local

    def render(self, origins, viewdirs, c2w, focal, width: int, height: int) -> torch.Tensor:
        return self.render_async(origins, viewdirs, c2w, focal, width, height)()

    def render_async(self, origins, viewdirs, c2w, focal, width: int, height: int):
        """Renders this rank's band and STARTS the gather; returns a callable that waits for it and returns the frame.
        A frame loop calls it after launching the next frame's band, so the exchange (RCCL's stream) overlaps that
        render.  Frames must be finished in the order they were started."""
        if origins.shape[0] != width * height:
            raise ValueError("origins / viewdirs must be the frame's full row-major ray arrays")
        cuts = self.last_cuts = self.cuts_for(height)
        self._band_samples = None
        rows = cuts[self.rank + 1] - cuts[self.rank]
        local = self.render_band(origins, viewdirs, c2w, focal, width, height, cuts[self.rank], cuts[self.rank + 1])
        meta = None
        if self.balance:                      # every rank sends its counts (zeros when its renderer reports none)
            meta = self._band_samples
            if meta is None or meta.numel() != rows:
                meta = torch.full((rows,), -1.0, dtype=torch.float32, device=local.device)
        pending = gather_bands(local, cuts, width, self.rank, self.world, async_op=True, meta=meta)

        def finish():
            if not self.balance:
                return pending()
            frame, row_samples = pending()
            self._push_profile(frame, width, height, row_samples)
            return frame

        return finish

    def _push_profile(self, frame, width, height, row_samples=None):
        """Per-row cost inputs of this frame -> pinned host memory; the profile of the frame rendered ``PROFILE_LAG``
        calls earlier becomes ``row_cost``.  Row 0 of the pair: pixels with alpha > 0; row 1: quadrature points
        (negative where a band's renderer reported none)."""
        obj = (frame[:, 3] > 0).view(height, width).sum(dim=1, dtype=torch.float32)
        smp = row_samples.to(torch.float32).reshape(-1) if row_samples is not None else torch.full_like(obj, -1.0)
        both = torch.stack([obj, smp])
        if both.is_cuda:
            ring = self._ring.get(height)
            if ring is None:      # PROFILE_LAG + 1 pinned buffers, reused round-robin (pinning per frame is slow)
                ring = self._ring[height] = [[torch.empty((2, height), dtype=torch.float32).pin_memory()
                                              for _ in range(self.PROFILE_LAG + 1)], 0]
            host = ring[0][ring[1] % len(ring[0])]
            ring[1] += 1
            host.copy_(both, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = both.clone(), None
        self._profiles.append((host, ev))
        if len(self._profiles) >= self.PROFILE_LAG:       # deterministic: the same frame's profile on every rank
            host, ev = self._profiles.pop(0)
            if ev is not None:
                ev.synchronize()
            obj_h, smp_h = host.numpy().astype(np.float64)
            cost = COST_RAY * width + np.where(smp_h >= 0, COST_SAMPLE * smp_h, COST_OBJECT * obj_h)
            self.row_cost = cost


# ----------------------------------------------------------------------------------------------------------
# 8x8 tiles dealt round-robin (arbitrary ray sets; no camera-coherent fast path)

def tile_layout(width: int, height: int) -> Tuple[int, int]:
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def shard_tiles(width: int, height: int, rank: int, world_size: int) -> torch.Tensor:
    """Tile ids owned by ``rank``; padded with -1 so every rank owns the same count."""
    tx, ty = tile_layout(width, height)
    n_tiles = tx * ty
    per_rank = (n_tiles + world_size - 1) // world_size
    ids = torch.arange(rank, rank + per_rank * world_size, world_size)
    ids[ids >= n_tiles] = -1
    return ids


def tile_ray_indices(tile_ids: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """[n_tiles, 64] ray ids (row-major pixel index) of each tile; -1 for pixels outside the image / pad tiles."""
    tx, _ = tile_layout(width, height)
    t = tile_ids.clamp_min(0)
    px = (t % tx)[:, None] * TILE + (torch.arange(TILE * TILE) % TILE)[None, :]
    py = (t // tx)[:, None] * TILE + (torch.arange(TILE * TILE) // TILE)[None, :]
    ray = py * width + px
    ray[(px >= width) | (py >= height) | (tile_ids[:, None] < 0)] = -1
    return ray


def local_rays(origins: torch.Tensor, viewdirs: torch.Tensor, width: int, height: int, rank: int, world_size: int):
    """This rank's rays in tile order: (origins_local, viewdirs_local, ray_ids [n_local]); pad slots repeat ray 0
    and are dropped again by ``gather_frame`` (their id is -1)."""
    ids = tile_ray_indices(shard_tiles(width, height, rank, world_size), width, height).reshape(-1)
    src = ids.clamp_min(0).to(origins.device)
    return origins[src].contiguous(), viewdirs[src].contiguous(), ids.to(origins.device)


def gather_frame(local: torch.Tensor, width: int, height: int, rank: int, world_size: int) -> torch.Tensor:
    """local [n_local, C] (this rank's tile-ordered results) -> full frame [H*W, C] on every rank, one collective."""
    c = local.shape[1]
    if world_size == 1:
        gathered = local[None]
    else:
        # concatenated layout [world*n_local, C]: accepted by both RCCL and gloo
        buf = torch.empty((world_size * local.shape[0], c), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, local.contiguous())
        gathered = buf.view(world_size, local.shape[0], c)
    frame = torch.zeros((width * height, c), dtype=local.dtype, device=local.device)
    for r in range(world_size):
        ids = tile_ray_indices(shard_tiles(width, height, r, world_size), width, height).reshape(-1).to(local.device)
        keep = ids >= 0
        frame[ids[keep]] = gathered[r][keep]
    return frame


# ----------------------------------------------------------------------------------------------------------
# whole frames dealt to the ranks (weak scaling: one frame per rank per step)

#: where the finished frames of a frame-parallel run go.  "rank0" (default): ``dist.gather`` to rank 0, the only rank
#: that scores or stores an evaluation run's images (train_finetune.py:575-629 runs on one device) -- the other ranks
#: receive nothing.  "all": ``all_gather_into_tensor``, every rank ends up with every frame (rounds 1-3; 7 x 12.8 MB
#: inbound per rank per 1.6 ms step at 8 ranks, beside a fabric-bound field kernel).  "none": no data-path collective.
FRAME_GATHER_MODE = "rank0"
FRAME_GATHER_MODES = ("rank0", "all", "none")
BAND_GATHER_MODES = ("exact", "padded", "broadcast")


def gather_frames(mine: torch.Tensor, rank: int, world_size: int, mode: Optional[str] = None,
                  out: Optional[torch.Tensor] = None, async_op: bool = False, force_collective: bool = False):
    """mine [n, C] (this rank's finished frame; the same shape on every rank) -> ``[world, n, C]`` on rank 0 ("rank0")
    or on every rank ("all"); ``None`` on the ranks that receive nothing.  ``out``: the receive buffer to use
    ([world * n, C], on the ranks that receive).  ``async_op``: returns a zero-argument callable that waits for the
    collective (current stream) and returns the result.  The gloo rehearsal on one GPU stages through host memory."""
    mode = FRAME_GATHER_MODE if mode is None else mode
    if mode not in FRAME_GATHER_MODES:
        raise ValueError(f"frame gather mode {mode!r}")
    n, c = mine.shape
    if mode == "none" or (world_size == 1 and not force_collective):
        result = None if mode == "none" else mine.view(1, n, c)
        return (lambda: result) if async_op else result
    staged = mine.is_cuda and dist.get_backend() == "gloo"
    send = (mine.cpu() if staged else mine).contiguous()
    receives = mode == "all" or rank == 0
    buf = None
    if receives:
        buf = out if out is not None else torch.empty((world_size * n, c), dtype=send.dtype, device=send.device)
        if buf.shape != (world_size * n, c) or buf.device != send.device:
            raise ValueError("gather_frames: out must be [world * n, C] on the collective's device")
    if mode == "all":
        work = dist.all_gather_into_tensor(buf, send, async_op=async_op)
    else:
        parts = list(buf.view(world_size, n, c).unbind(0)) if rank == 0 else None
        work = dist.gather(send, gather_list=parts, dst=0, async_op=async_op)

    def finish():
        if work is not None:
            work.wait()                    # send / buf stay referenced by this closure until then
        if buf is None:
            return None
        res = buf.view(world_size, n, c)
        return res.to(mine.device) if staged else res

    return finish if async_op else finish()


# ----------------------------------------------------------------------------------------------------------
# the first thing an N > 1 run does: prove the collectives before anything is timed

class CollectiveSelfTestError(RuntimeError):
    """No usable collective for a data-path exchange (or the ranks' plain all_reduce failed): the run must stop."""


def _selftest_band(r: int, rows: int, width: int, c: int) -> torch.Tensor:
    """The analytically known band of rank r: value = 1000 r + 10 (row-major ray index) + channel (exact in fp32)."""
    ray = torch.arange(rows * width, dtype=torch.float32)[:, None]
    return 1000.0 * r + 10.0 * ray + torch.arange(c, dtype=torch.float32)[None, :]


def device_report(rank: int, world_size: int, device: Optional[torch.device] = None) -> dict:
    """Which device every rank computes on, gathered with the group's plain all_reduce (int64 SUM of a one-hot table):
    ``torch.cuda.current_device()``, PCI domain / bus / device and a 62-bit digest of the device UUID per rank, and the
    number of DISTINCT physical devices among them (must equal the world size on a real N-GPU run; 1 in the one-GPU
    rehearsal)."""
    row = [-1, -1, -1, -1, 0]
    if device is not None and device.type == "cuda":
        import hashlib
        idx = torch.cuda.current_device()
        props = torch.cuda.get_device_properties(idx)
        uuid = str(getattr(props, "uuid", ""))
        row = [idx, int(getattr(props, "pci_domain_id", -1)), int(getattr(props, "pci_bus_id", -1)),
               int(getattr(props, "pci_device_id", -1)),
               int.from_bytes(hashlib.sha256(uuid.encode()).digest()[:8], "little") >> 2 if uuid else 0]
    on = device if (device is not None and dist.is_initialized() and dist.get_backend() == "nccl") else torch.device("cpu")
    table = torch.zeros((world_size, len(row)), dtype=torch.int64, device=on)
    table[rank] = torch.tensor(row, dtype=torch.int64, device=on)
    if world_size > 1 or (dist.is_available() and dist.is_initialized()):
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
    rows = table.cpu().tolist()
    ranks = [{"rank": r, "cuda_device": v[0], "pci": (f"{v[1]:04x}:{v[2]:02x}:{v[3]:02x}" if v[2] >= 0 else None),
              "uuid_digest": f"{v[4]:016x}" if v[4] else None} for r, v in enumerate(rows)]
    # physical identity: PCI address when the runtime reports one, else the UUID digest, else the device ordinal
    ident = [(v[1], v[2], v[3]) if v[2] >= 0 else (("uuid", v[4]) if v[4] else ("ordinal", v[0])) for v in rows]
    return {"ranks": ranks, "distinct_devices": len(set(ident))}


def selftest_collectives(rank: int, world_size: int, device: Optional[torch.device] = None,
                         band_modes: Sequence[str] = BAND_GATHER_MODES,
                         frame_modes: Sequence[str] = ("rank0", "all"), select: bool = True) -> dict:
    """Runs every data-path collective of this module once on ~1 KB of analytically known data -- on ``device``
    tensors (RCCL) or host tensors (gloo) -- BEFORE anything is timed, and picks the modes to use:

    * band exchange (``gather_bands``): uneven bands with their per-row numbers riding along, blocking and async, for
      every mode of ``band_modes`` in order of preference (all_to_all_single with split sizes, the padded
      all_gather_into_tensor, per-rank broadcasts);
    * frame exchange (``gather_frames``): ``dist.gather`` to rank 0, then ``all_gather_into_tensor``;
    * the float64 MAX / SUM all_reduce the benchmark takes its max-over-ranks time with.

    A mode counts only if EVERY rank got the right bytes (the ranks agree through an int32 MIN all_reduce).  With
    ``select`` the first good mode of each kind becomes ``GATHER_MODE`` / ``FRAME_GATHER_MODE``.  Raises
    ``CollectiveSelfTestError`` carrying the backend's error text when the plain all_reduce fails or no mode of a kind
    works -- a wrong frame must never be timed.  The returned dict goes into bench.py's JSON line."""
    global GATHER_MODE, FRAME_GATHER_MODE
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": world_size, "band_gather_mode": GATHER_MODE,
                "frame_gather_mode": FRAME_GATHER_MODE, "tested": {}, "note": "no process group (one rank)"}
    backend = dist.get_backend()
    on = device if (backend == "nccl" and device is not None) else torch.device("cpu")
    data_dev = device if device is not None else torch.device("cpu")     # where a rank's results live
    if dist.get_world_size() != world_size or dist.get_rank() != rank:
        raise CollectiveSelfTestError(f"process group says rank {dist.get_rank()} of {dist.get_world_size()}, the "
                                      f"environment rank {rank} of {world_size}")
    tested = {}
    # (0) the plain reductions everything else leans on
    try:
        t = torch.tensor([float(rank + 1), 1.0], dtype=torch.float64, device=on)
        tmax, tsum = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        want_max, want_sum = [float(world_size), 1.0], [world_size * (world_size + 1) / 2.0, float(world_size)]
        if tmax.cpu().tolist() != want_max or tsum.cpu().tolist() != want_sum:
            raise CollectiveSelfTestError(f"all_reduce gave MAX {tmax.cpu().tolist()} SUM {tsum.cpu().tolist()}, "
                                          f"expected {want_max} / {want_sum}")
        tested["all_reduce_f64"] = "ok"
    except CollectiveSelfTestError:
        raise
    except Exception as e:                                              # noqa: BLE001 -- the backend's own error text
        raise CollectiveSelfTestError(f"{backend} all_reduce failed on rank {rank}: {type(e).__name__}: {e}") from e

    def agree(flags):
        f = torch.tensor(flags, dtype=torch.int32, device=on)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return [bool(v) for v in f.cpu().tolist()]

    # (1) band exchange: rank r owns (r % 3) + 1 rows of a 3-wide frame, 5 channels; per-row numbers = global row index
    width, c = 3, 5
    rows = [(r % 3) + 1 for r in range(world_size)]
    cuts = [sum(rows[:r]) for r in range(world_size + 1)]
    want_frame = torch.cat([_selftest_band(r, rows[r], width, c) for r in range(world_size)], dim=0)
    want_meta = torch.arange(cuts[-1], dtype=torch.float32)
    local = _selftest_band(rank, rows[rank], width, c).to(data_dev)
    meta = torch.arange(cuts[rank], cuts[rank + 1], dtype=torch.float32, device=data_dev)
    errors, flags = {}, []
    for mode in band_modes:
        try:
            frame, m = gather_bands(local, cuts, width, rank, world_size, meta=meta, mode=mode, force_collective=True)
            frame2 = gather_bands(local, cuts, width, rank, world_size, mode=mode, async_op=True, force_collective=True)()
            ok = (torch.equal(frame.cpu(), want_frame) and torch.equal(m.cpu(), want_meta)
                  and torch.equal(frame2.cpu(), want_frame))
            errors[mode] = None if ok else "wrong bytes"
        except Exception as e:                                          # noqa: BLE001
            ok, errors[mode] = False, f"{type(e).__name__}: {e}"
        flags.append(int(ok))
    good = agree(flags)
    for mode, g in zip(band_modes, good):
        tested[f"band:{mode}"] = "ok" if g else (errors[mode] or "failed on another rank")
    band_choice = next((m for m, g in zip(band_modes, good) if g), None)

    # (2) frame exchange: every rank sends 6 x 5 numbers
    mine = _selftest_band(rank, 2, width, c).to(data_dev)
    want_all = torch.stack([_selftest_band(r, 2, width, c) for r in range(world_size)])
    errors, flags = {}, []
    for mode in frame_modes:
        try:
            got = gather_frames(mine, rank, world_size, mode=mode, force_collective=True)
            got2 = gather_frames(mine, rank, world_size, mode=mode, async_op=True, force_collective=True)()
            if mode == "rank0" and rank != 0:
                ok = got is None and got2 is None
            else:
                ok = torch.equal(got.cpu(), want_all) and torch.equal(got2.cpu(), want_all)
            errors[mode] = None if ok else "wrong bytes"
        except Exception as e:                                          # noqa: BLE001
            ok, errors[mode] = False, f"{type(e).__name__}: {e}"
        flags.append(int(ok))
    good = agree(flags)
    for mode, g in zip(frame_modes, good):
        tested[f"frame:{mode}"] = "ok" if g else (errors[mode] or "failed on another rank")
    frame_choice = next((m for m, g in zip(frame_modes, good) if g), None)

    if band_choice is None or frame_choice is None:
        raise CollectiveSelfTestError(f"{backend}, {world_size} ranks: no working collective for the "
                                      f"{'band' if band_choice is None else 'frame'} exchange: {tested}")
    if select:
        if GATHER_MODE not in band_modes or tested.get(f"band:{GATHER_MODE}") != "ok":
            GATHER_MODE = band_choice
        if FRAME_GATHER_MODE != "none" and tested.get(f"frame:{FRAME_GATHER_MODE}") != "ok":
            FRAME_GATHER_MODE = frame_choice
    return {"backend": backend, "world_size": dist.get_world_size(), "band_gather_mode": GATHER_MODE,
            "frame_gather_mode": FRAME_GATHER_MODE, "tested": tested}


# ----------------------------------------------------------------------------------------------------------

def init_from_env(backend: str = "nccl"):
    """(rank, local_rank, world_size) from the torchrun environment; initialises the process group if needed."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a collective that never completes should fail the run in minutes, with a message, not sit until the launcher's
        # own limit (the default is 10 min for RCCL, 30 for gloo); everything exchanged here is a few MB
        limit = datetime.timedelta(seconds=float(os.environ.get("QF_COLLECTIVE_TIMEOUT_S", "300")))
        if backend == "nccl" and torch.cuda.is_available():
            # bind the rank to its GPU before RCCL comes up, and tell the process group which device it owns
            torch.cuda.set_device(local_rank)
            try:
                dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=limit,
                                        device_id=torch.device("cuda", local_rank))
            except TypeError:               # a torch without the device_id keyword: lazy communicator, same binding
                dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=limit)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=limit)
    return rank, local_rank, world


def shutdown() -> None:
    """Tear the process group down (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
