"""Ray-batch sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in CPU tests).

The reference is single-GPU; this is new.  Rays are independent, so a frame is cut
into 8x8-pixel tiles dealt round-robin to ranks (object and background tiles balance), every rank holds a
replica of mesh/BVH + tables + weights, renders its tiles with no collective on the data path, and the
finished tiles (rgb3 + alpha1 + depth1 = 20 B/ray) are exchanged with ONE all_gather_into_tensor per frame.
At 800x800 that is 12.8 MB in total, 1.6 MB per rank on 8 ranks: latency-, not bandwidth-bound on xGMI.
"""
from typing import Tuple

import torch
import torch.distributed as dist

TILE = 8


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
    and are dropped again by ``scatter_tiles`` (their id is -1)."""
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


def init_from_env(backend: str = "nccl"):
    """(rank, local_rank, world_size) from the torchrun environment; initialises the process group if needed."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl" and torch.cuda.is_available():
            # bind the rank to its GPU before RCCL comes up, and tell the process group which device it owns
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shutdown() -> None:
    """Tear the process group down (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
