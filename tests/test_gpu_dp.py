"""Two data-parallel ranks sharing one MI355X (gloo rendezvous on 127.0.0.1, both on cuda:0):
the real kernels under the gradient exchange, the deferred critic step and the packed-weight
cache. After a few loop bodies both ranks must hold identical parameters (same initial weights,
averaged gradients) and finite losses, in eager and in captured-graph mode."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, graphs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    gen, critic = bench.build_models(dev, 120)
    eng = Phase3Engine(gen, critic, dict(bench.P3_DEFAULT, n_critic_steps=2))
    assert eng.x_critic is not None and eng.x_critic.active
    if graphs:
        eng.enable_graphs()
    real, audio, slices = synthetic_phase3_batch(4, 120, dev, seed=50 + rank)  # different shard per rank
    torch.manual_seed(7)  # same host draws on both ranks is fine: the shards differ
    for _ in range(4):
        out = eng.train_step(real, audio, slices)
    eng.flush()
    torch.cuda.synchronize()
    vals = {k: float(v) for k, v in out.items()}
    sig = [float(p.detach().double().sum()) for p in list(critic.parameters())[:6] + list(gen.parameters())[:6]]
    q.put((rank, vals, sig))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("graphs", [False, True], ids=["eager", "graphs"])
def test_two_ranks_one_gpu(graphs):
    import math
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, graphs, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=300) for _ in range(world))
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, v0, s0), (_, v1, s1) = res
    assert all(math.isfinite(x) for x in list(v0.values()) + list(v1.values()))
    assert set(v0) == {"loss_critic", "gp", "w_dist", "loss_gen", "l1_loss_train"}
    for a, b in zip(s0, s1):  # parameters stay in lock-step across ranks
        assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (s0, s1)
