"""Multi-GPU sharding of independent PTZ problems.

Scenes (run_ptzba_synthetic.sh:4-13 runs one process per scene) and relocalization queries
(run_ptz_reloc.cc:68 loop) never interact, so the path shards at problem granularity: one process per GPU,
a static block partition of the work items, no collective on the data path.  torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only to gather the small result
blocks (15 doubles per camera + a summary per problem) on rank 0.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Static block partition: items [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return range(lo, hi)


def greedy_partition(costs, world: int):
    """Longest-first greedy assignment for heterogeneous problems (e.g. WorldCup14 matches of different
    size).  Returns a list of item-index lists, one per rank; deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    for r in range(world):
        out[r].sort()
    return out


def gather_results(local_ids, local_payload: np.ndarray, n_items: int, dist=None, device=None):
    """Gather per-item fixed-width float64 payload rows on every rank (all_gather of a padded block).

    local_ids: item indices owned by this rank; local_payload: [len(local_ids), width].
    Returns [n_items, width] on every rank.  With dist=None (single process) it is a local scatter."""
    width = local_payload.shape[1] if local_payload.ndim == 2 else 0
    out = np.zeros((n_items, width))
    if dist is None or not dist.is_initialized():  # (a one-rank group still goes through the collective)
        out[list(local_ids)] = local_payload
        return out
    import torch

    world = dist.get_world_size()
    cap = (n_items + world - 1) // world + 1
    blk = torch.zeros((cap, width + 1), dtype=torch.float64)
    blk[:, 0] = -1
    ids = np.asarray(list(local_ids), dtype=np.float64)
    if len(ids):
        blk[: len(ids), 0] = torch.from_numpy(ids)
        blk[: len(ids), 1:] = torch.from_numpy(np.ascontiguousarray(local_payload, dtype=np.float64).reshape(len(ids), width))
    if device is not None:
        blk = blk.to(device)
    parts = [torch.empty_like(blk) for _ in range(world)]
    dist.all_gather(parts, blk)
    for p in parts:
        p = p.cpu().numpy()
        rows = p[:, 0] >= 0
        out[p[rows, 0].astype(np.int64)] = p[rows, 1:]
    return out


def solve_scenes_sharded(scene_ids, make_scene, solve_batch, dist=None, device=None, cam_width=None):
    """Shard `scene_ids` over the ranks, solve the local shard with `solve_batch(scenes) -> (cams, summaries)`
    and gather [cameras | termination, iterations, final cost] per scene on every rank."""
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    mine = [scene_ids[i] for i in shard_range(len(scene_ids), rank, world)]
    scenes = [make_scene(s) for s in mine]
    cams, summ = solve_batch(scenes) if scenes else ([], [])
    n_cam = cam_width if cam_width is not None else (scenes[0].n_cam if scenes else 0)
    width = 15 * n_cam + 3
    payload = np.zeros((len(mine), width))
    for k in range(len(mine)):
        payload[k, : 15 * scenes[k].n_cam] = np.asarray(cams[k]).reshape(-1)
        payload[k, -3:] = [summ[k]["termination_type"], summ[k]["num_iterations"], summ[k]["final_cost"]]
    local_idx = list(shard_range(len(scene_ids), rank, world))
    return gather_results(local_idx, payload, len(scene_ids), dist, device)
