"""Independent MCMC chains across GPUs: chains are replicas, one engine per
process per GPU, no data-path collective.  torch.distributed is used only for the barrier
and for gathering host scalars (timings, per-chain log-likelihoods): host data, so the process
group is gloo at every rank count -- nothing of the job depends on RCCL (SBAYES_AMD_DIST_BACKEND=nccl
opts into it; tests/test_gpu_dist_backend.py runs that branch once on hardware).  This mirrors the reference's MC3 layout (one OS process per chain exchanging
host scalars: sbayes/mcmc_setup.py:271-299, 386-409)."""
from __future__ import annotations

import os


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_chains(n_chains: int, world_size: int, rank: int):
    """Contiguous block partition of chain ids over ranks (sizes differ by at most one)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(n_chains, world_size)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return list(range(lo, hi))


def device_for(local_rank: int, n_devices: int) -> int:
    if n_devices < 1:
        raise RuntimeError("no GPU visible: the engine has no CPU fallback")
    return local_rank % n_devices


def init_process_group(backend=None):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1; returns the module or None."""
    _, local_rank, world = env_rank()
    if world <= 1 and not (os.environ.get("SBAYES_AMD_FORCE_DIST") and "MASTER_ADDR" in os.environ):
        return None
    import torch
    import torch.distributed as dist
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if backend is None:
        # barrier + one float64 max + an object gather of host scalars: gloo serves them at every rank count; the data
        # path has no collective, so depending on RCCL would be a risk without a benefit
        backend = os.environ.get("SBAYES_AMD_DIST_BACKEND") or "gloo"
    if backend == "nccl":
        if n_dev < 1:
            raise RuntimeError("SBAYES_AMD_DIST_BACKEND=nccl without a visible GPU")
        if n_dev < local_world:
            print(f"[chains] nccl with {local_world} local ranks on {n_dev} device(s): ranks share devices", flush=True)
        torch.cuda.set_device(local_rank % n_dev)
    if not dist.is_initialized():
        try:
            dist.init_process_group(backend=backend)
        except Exception as exc:          # the data path has no collective: gloo serves the barrier equally
            if backend == "gloo":
                raise
            print(f"[chains] {backend} init failed ({exc}); falling back to gloo", flush=True)
            dist.init_process_group(backend="gloo")
            dist._sbayes_amd_note = f"gloo (after {backend} failed to initialise)"
    return dist


def backend_name(dist) -> str:
    if dist is None:
        return "none (single process)"
    return getattr(dist, "_sbayes_amd_note", None) or str(dist.get_backend())


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(value: float, dist) -> float:
    if dist is None:
        return value
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_chain_values(local_ids, local_values, n_chains: int, dist):
    """All ranks receive the per-chain host scalars (e.g. log-likelihoods for an MC3 swap
    decision) ordered by chain id."""
    import numpy as np
    out = np.full(n_chains, np.nan)
    if dist is None:
        out[list(local_ids)] = local_values
        return out
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, (list(local_ids), [float(v) for v in local_values]))
    for ids, vals in gathered:
        out[ids] = vals
    return out
