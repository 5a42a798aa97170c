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


def _watchdog(name, seconds=240):
    """A stuck worker dumps every thread's Python stack (gpurun_out/ travels back from the GPU box) and exits, so the
    parent fails in minutes instead of waiting out its queue."""
    import faulthandler
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "stuck_%s_%d.log" % (name, os.getpid()))
    f = open(path, "w")
    faulthandler.dump_traceback_later(seconds, exit=True, file=f)

    def _done():  # a worker that finished leaves no file
        faulthandler.cancel_dump_traceback_later()
        f.close()
        if os.path.getsize(path) == 0:
            os.unlink(path)
    import atexit
    atexit.register(_done)
    return f


class _Stuck(AssertionError):
    """A worker stuck (watchdog) or died without a result. NOT retried (round-3 advice): this code hands data between
    workgroups with spin loops, zero-kept counters and completion-ordered streams, so an unexplained hang is a finding,
    not a flake - the run fails and gpurun_out/stuck_*.log holds every thread's stack. Levers for the post-mortem:
    M2D_FUSED_SPLITK=0, M2D_PERSISTENT_GRU=0, M2D_MANUAL_CRITIC=0."""


def _recv(q, procs, n, timeout):
    """n results from the workers; fails as soon as a worker has died without reporting."""
    import queue, time
    out, t0 = [], time.time()
    while len(out) < n:
        try:
            out.append(q.get(timeout=5))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or all(p.exitcode is not None for p in procs):
                raise _Stuck("worker exited without a result: exit codes %s" % [p.exitcode for p in procs])
            if time.time() - t0 > timeout:
                [p.kill() for p in procs]
                raise _Stuck("workers timed out after %d s" % timeout)
    return out


def _worker(rank, world, port, graphs, q):
    _wd = _watchdog("two_ranks_r%d" % rank)
    os.environ["M2D_PERSISTENT_GRU"] = "0"  # two processes on ONE GPU: persistent kernels could starve each other
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
    res = sorted(_recv(q, procs, world, 300))
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, v0, s0), (_, v1, s1) = res
    assert all(math.isfinite(x) for x in list(v0.values()) + list(v1.values()))
    assert set(v0) == {"loss_critic", "gp", "w_dist", "loss_gen", "l1_loss_train"}
    for a, b in zip(s0, s1):  # parameters stay in lock-step across ranks
        assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (s0, s1)


def _rccl_worker(port, q):
    """backend "nccl" = RCCL: the exchange (fused pack, ReduceOp.AVG on the communication stream, gradient
    views) and the deferred critic step on the real library, at world size 1 (one GPU per box here)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _wd = _watchdog("rccl")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import bench
    from music2dance_amd.dp import GradExchange
    from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
    dev = torch.device("cuda:0")
    # the exchange alone: averaged over one rank = unchanged values, now views of the buckets
    params = [torch.nn.Parameter(torch.randn(300, 7, device=dev)), torch.nn.Parameter(torch.randn(11, device=dev)),
              torch.nn.Parameter(torch.randn(5, device=dev))]
    params[0].grad, params[1].grad = torch.randn(300, 7, device=dev), torch.randn(11, device=dev)
    want = [params[0].grad.clone(), params[1].grad.clone()]
    ex = GradExchange(params, bucket_mb=1e-3, force=True)
    assert ex.active and len(ex.buckets) >= 2
    ex.exchange()
    torch.cuda.synchronize()
    ok = torch.equal(params[0].grad, want[0]) and torch.equal(params[1].grad, want[1]) and params[2].grad is None
    flat_ptrs = {v.data_ptr() for views in ex._views for v in views}
    ok = ok and params[0].grad.data_ptr() in flat_ptrs and params[1].grad.data_ptr() in flat_ptrs
    # the engine with the exchange forced on: same losses and parameters as without it
    sigs = []
    for forced in (False, True):
        gen, critic = bench.build_models(dev, 120)
        eng = Phase3Engine(gen, critic, dict(bench.P3_DEFAULT, n_critic_steps=2))
        eng.x_critic.force = eng.x_gen.force = forced
        real, audio, slices = synthetic_phase3_batch(4, 120, dev, seed=60)
        torch.manual_seed(9)
        for _ in range(4):
            out = eng.train_step(real, audio, slices)
        eng.flush()
        torch.cuda.synchronize()
        sigs.append(([float(v) for v in out.values()],
                     [float(p.detach().double().sum()) for p in list(critic.parameters())[:6] + list(gen.parameters())[:6]],
                     int(next(iter(eng.optim_critic.state.values()))["step"])))
        # from the second iteration on the critic's buckets leave from the backward hooks (pose-branch gradients
        # are accumulated on the side stream: the launch waits for every stream's mark)
        ok = ok and (eng.x_critic.launched_in_backward >= 3 * len(eng.x_critic.buckets) if forced
                     else eng.x_critic.launched_in_backward == 0)
    q.put((ok, sigs))
    dist.destroy_process_group()


def test_rccl_backend_exchange_world_size_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    ok, sigs = _recv(q, [p], 1, 300)[0]
    p.join(120)
    assert p.exitcode == 0 and ok
    (l0, s0, n0), (l1, s1, n1) = sigs
    assert n0 == n1 == 4  # every critic step taken, deferred or not
    for a, b in zip(l0 + s0, l1 + s1):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (sigs)


def _rccl_graphs_worker(port, q):
    """RCCL (world size 1, exchange forced on) with everything that meets on a real 8-GPU run in ONE step: the
    persistent GRU launch, the generator forward pipelined on its own stream (inputs_ready), the critic's buckets
    leaving from backward hooks - and, in graph mode, a SECOND batch shape captured after eager exchanges have armed
    the hooks (the warm-up backward of that capture must not feed the exchange: GradExchange.suspended)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert os.environ.get("M2D_PERSISTENT_GRU", "1") != "0"
    _wd = _watchdog("rccl_graphs")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    import bench
    from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch
    dev = torch.device("cuda:0")
    batches = {b: synthetic_phase3_batch(b, 120, dev, seed=70 + b, with_event=True) for b in (4, 2)}
    order = (4, 4, 2, 4, 2, 2, 4, 4)
    sigs = []
    for graphs in (False, True):
        gen, critic = bench.build_models(dev, 120)
        eng = Phase3Engine(gen, critic, dict(bench.P3_DEFAULT, n_critic_steps=2))
        eng.x_critic.force = eng.x_gen.force = True
        if graphs:
            eng.enable_graphs()
        torch.manual_seed(11)
        for b in order:
            real, audio, slices, ready = batches[b]
            out = eng.train_step(real, audio, slices, inputs_ready=ready)
        eng.flush()
        torch.cuda.synchronize()
        eng._check_async()
        sigs.append(([float(v) for v in out.values()],
                     [float(p.detach().double().sum()) for p in list(critic.parameters()) + list(gen.parameters())],
                     int(next(iter(eng.optim_critic.state.values()))["step"]), eng.x_critic.launched_in_backward))
    q.put(sigs)
    dist.destroy_process_group()


def test_rccl_two_shapes_graphs_equal_eager_with_persistent_gru_and_hook_overlap():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_graphs_worker, args=(_free_port(), q))
    p.start()
    sigs = _recv(q, [p], 1, 300)[0]
    p.join(120)
    assert p.exitcode == 0
    (l0, s0, n0, h0), (l1, s1, n1, h1) = sigs
    assert n0 == n1 == 8
    assert h0 > 0  # eager: buckets left from the backward hooks (captured backward passes keep the exchange outside)
    for a, b in zip(l0 + s0, l1 + s1):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), sigs
