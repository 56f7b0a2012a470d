"""N>1 path on CPU: two gloo ranks shard the chains, meet at the barrier, agree on the
max-over-ranks time and exchange per-chain host scalars -- the only traffic the multi-GPU
layout has (no data-path collective)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from sbayes_amd import chains


def test_shard_chains_partitions_exactly():
    for n_chains in (1, 2, 7, 8, 64):
        for world in (1, 2, 3, 8):
            parts = [chains.shard_chains(n_chains, world, r) for r in range(world)]
            flat = [c for p in parts for c in p]
            assert flat == list(range(n_chains))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        chains.shard_chains(4, 2, 2)
    assert chains.device_for(5, 8) == 5 and chains.device_for(9, 8) == 1
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        chains.device_for(0, 0)


def _worker(rank, world, port, n_chains, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = chains.init_process_group(backend="gloo")
    assert chains.env_rank() == (rank, rank, world)
    mine = chains.shard_chains(n_chains, world, rank)
    values = [-100.0 - 3.0 * c for c in mine]            # stand-in for per-chain log-likelihoods
    chains.barrier(dist)
    slowest = chains.max_over_ranks(1.0 + rank, dist)
    allv = chains.gather_chain_values(mine, values, n_chains, dist)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.concatenate([[slowest], allv]))
    chains.barrier(dist)
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n_chains, world = 5, 2
    mp.spawn(_worker, args=(world, port, n_chains, str(tmp_path)), nprocs=world, join=True)
    want = np.array([-100.0 - 3.0 * c for c in range(n_chains)])
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got[0] == 2.0                      # max over ranks of (1 + rank)
        assert np.array_equal(got[1:], want)


def test_single_process_path_needs_no_dist():
    assert chains.max_over_ranks(3.5, None) == 3.5
    assert np.array_equal(chains.gather_chain_values([0, 1], [1.0, 2.0], 2, None), [1.0, 2.0])
