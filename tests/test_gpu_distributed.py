"""N > 1 path with the REAL solver: two ranks, each driving its shard of the global
batch through BatchILQR on the one GPU of the test box (gloo carries the gather; on an 8-GPU node
bench.py does the same with one GPU per rank over RCCL), must return exactly -- bit for bit -- what one
process returns for the whole batch: trajectories never interact, the shard is a contiguous block and
the gather is rank-ordered."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import acrobot_x0

pytestmark = pytest.mark.gpu
DT, T, LIM, ITERS = 0.02, 90, 1.5, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _solve(x0):
    from ilqr_amd import BatchILQR
    B = len(x0)
    g = BatchILQR("acrobot", B, T, DT, u_min=-LIM, u_max=LIM, device=0)
    g.init_traj(x0, np.zeros((B, T, 1)))
    g.iterate(ITERS)
    st, it, al = g.status()
    out = (g.cost(), st.copy(), it.copy())
    g.close()
    return out


def _worker(rank, ws, port, B, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from ilqr_amd import dist as D
    lo, hi = D.shard(B * ws, rank, ws)
    cost, st, it = _solve(acrobot_x0(B * ws)[lo:hi])
    allc = D.gather_costs(torch.from_numpy(cost))
    alls = D.gather_costs(torch.from_numpy(st.astype(np.int64)))
    D.barrier()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), cost=allc.numpy(), status=alls.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [48, 40])  # 40: a shard that is not a whole number of 16-trajectory tiles
def test_two_ranks_equal_one_process(tmp_path, B):
    ws = 2
    mp.spawn(_worker, args=(ws, _free_port(), B, str(tmp_path)), nprocs=ws, join=True)
    cost, st, it = _solve(acrobot_x0(B * ws))
    assert np.all(np.isfinite(cost))
    for r in range(ws):
        got = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(got["cost"], cost)  # bit-identical, in global order, on every rank
        assert np.array_equal(got["status"], st)
