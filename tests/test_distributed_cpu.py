"""N > 1 path on CPU: world_size-2 gloo run of the shard arithmetic, the gather of per-trajectory
costs and the max-over-ranks timing that bench.py uses (the data path itself needs no collective)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import acrobot_x0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from ilqr_amd import dist as D
    lo, hi = D.shard(B * ws, rank, ws)
    assert (lo, hi) == (rank * B, (rank + 1) * B)
    x0 = acrobot_x0(B * ws)[lo:hi]  # this rank's slice of the global synthetic batch
    # stand-in for the solver's result: a deterministic function of the trajectory's own x0 only
    local = torch.from_numpy(np.sum(x0 * x0, axis=1) + 1000.0 * np.arange(lo, hi))
    allc = D.gather_costs(local)
    t = D.max_over_ranks(0.5 + rank)
    D.barrier()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), allc.numpy())
    assert t == 0.5 + (ws - 1)
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process(tmp_path):
    ws, B = 2, 48
    port = _free_port()
    mp.spawn(_worker, args=(ws, port, B, str(tmp_path)), nprocs=ws, join=True)
    x0 = acrobot_x0(B * ws)
    expect = np.sum(x0 * x0, axis=1) + 1000.0 * np.arange(B * ws)
    for r in range(ws):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert np.array_equal(got, expect)  # bit-identical to the unsharded order on every rank


def test_shard_rejects_ragged_batches():
    from ilqr_amd import dist as D
    import pytest
    with pytest.raises(ValueError):
        D.shard(10, 0, 4)
    assert D.shard(12, 3, 4) == (9, 12)
    assert D.world() == (0, 1)
    assert D.max_over_ranks(1.25) == 1.25
