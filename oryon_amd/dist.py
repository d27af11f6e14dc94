"""Multi-GPU layout of the path: pairs are independent (the reference loops `for i_b in range(BS)`,
pipeline.py:313), so a batch is block-partitioned over ranks with NO data-path collective; one
all_gather of [B_r, 17] fp32 rows (16 pose values + status) collates the result.
The reference itself has no collation step (each rank writes its own CSV, pipeline.py:480-484).

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm (xGMI), "gloo" on CPU for tests."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def warm_engine_streams(local: int) -> bool:
    """Create the step engine's stream pool on GPU `local` NOW, before anything else of the process creates HIP streams there.

    Which hardware queue an engine stream sits on moves the pipelined step by up to 30 % (DESIGN.md "stream placement"), the HIP runtime
    hands queues out in stream-creation order, and `oryon_engine_config_t.stream_roles` names positions in that order.  RCCL's
    communicator (eager with `device_id=`) creates streams of its own: created first, they would shift every role, and the N > 1 run
    would not have the placement the N = 1 number was tuned on.  So: pool first, process group second - at every N."""
    from ._lib import check, lib
    torch.cuda.set_device(local)
    check(lib().oryon_engine_warm_streams(), "oryon_engine_warm_streams")
    return True


def init_from_env(device_type: str = "cuda", force_group: bool = False) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank).  A single process without those variables is world 1 and needs no group;
    `force_group` creates one all the same (a one-rank RCCL communicator: the collation then runs through the same collectives as at
    N > 1 - what a single-GPU box can execute of the multi-GPU path, `bench.py --process-group`).
    On GPUs the step engine's stream pool is created BEFORE the process group (see warm_engine_streams), at world 1 too, so that a
    rank of an N-GPU run and a single-GPU run give their engine streams the same hardware queues."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type == "cuda":
        warm_engine_streams(local)
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sock:                  # (only a process that was not launched by torchrun gets here: it is alone)
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1]) if world == 1 else "29500"
        if device_type == "cuda":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, world, local


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r owns [start, stop) with ceil(total/world) items per rank (last ones short)."""
    per = (total + world - 1) // world
    start = min(rank * per, total)
    return start, min(start + per, total)


def gather_poses(pose_local: torch.Tensor, status_local: torch.Tensor, total: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Collate per-rank results into the global order.  pose_local [B_r,4,4] fp32, status_local [B_r] int32 ->
    (pose [total,4,4], status [total]) on every rank.  Ranks may hold fewer than ceil(total/world) pairs: rows are
    padded to equal size for the collective and cut afterwards."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if not dist.is_initialized():
        return pose_local[:total], status_local[:total]
    per = (total + world - 1) // world
    dev = pose_local.device
    packed = torch.zeros((per, 17), dtype=torch.float32, device=dev)
    b = pose_local.shape[0]
    packed[:b, :16] = pose_local.reshape(b, 16)
    packed[:b, 16] = status_local.to(torch.float32)
    out = torch.empty((world * per, 17), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, packed)
    out = out[:total]
    return out[:, :16].reshape(total, 4, 4).contiguous(), out[:, 16].to(torch.int32)


def gather_pose_windows(staged_local: torch.Tensor) -> torch.Tensor:
    """Final collation of a whole window (north_star: "all-gather of per-pair poses ... only for the final collation"): every rank staged
    the [B_r, 17] rows (16 pose values + status) of each of its k steps in `staged_local` [k, B_r, 17] fp32; ONE all_gather returns
    [world, k, B_r, 17] on every rank - step j of the job in global pair order is out[:, j].reshape(world * B_r, 17).
    All ranks must pass the same k and B_r (pad short ranks with status rows the reader cuts)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if not dist.is_initialized():
        return staged_local.unsqueeze(0)
    mine = staged_local.contiguous()
    out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out.view((world * mine.shape[0],) + tuple(mine.shape[1:])), mine)
    return out
