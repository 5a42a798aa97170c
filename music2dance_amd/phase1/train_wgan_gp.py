"""Phase 1: still-pose residual-MLP WGAN-GP on MI355X.

    python -m music2dance_amd.phase1.train_wgan_gp -c music2dance_amd/phase1/configs/b1l10s128.yaml -d 0 -n run --synthetic

Flags -d/-n as in the reference's phase1/train_wgan-gp.py plus -c for the config file (the
reference reads an undefined `file` variable, phase1/train_wgan-gp.py:31). The hyphenated
script name of the reference is not importable as a module; `train_wgan-gp.py` next to this
file forwards to it.
"""
import argparse

import numpy as np
import torch

from .. import dp, runner
from ..engine import Phase1Engine
from .archis.residual import Discriminator, Generator


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, help="choose config file")
    ap.add_argument("-d", "--device", type=int, help="choose gpu id")
    ap.add_argument("-n", "--name", type=str, help="choose name of experiment")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--folder", type=str, default=None, help="dataset folder (overrides the YAML's `folder:`)")
    ap.add_argument("--iterations", type=int, default=None)
    ap.add_argument("--log-every", type=int, default=1)
    ap.add_argument("--no-run-dir", action="store_true")
    ap.add_argument("--sync-bn", action="store_true", help="data parallel: BatchNorm statistics over the global batch")
    ap.add_argument("--host-loader", action="store_true",
                    help="fetch and collate batches on the host (torch DataLoader) instead of gathering them from the "
                         "HBM-resident dataset; same batches either way")
    ap.add_argument("--graphs", action="store_true",
                    help="replay each loop body from captured HIP graphs (the eager loop is bound by the host's launch "
                         "rate: 2.8 -> 0.86 ms per body at batch 64); results equal the eager path's. Default on one GPU")
    ap.add_argument("--no-graphs", action="store_true", help="keep the eager launch loop on a single GPU too")
    opts = ap.parse_args(argv)

    rank, world, local = dp.init_from_env()
    device = runner.pick_device(local if world > 1 else opts.device)
    cfg = runner.load_config(opts.config)
    loader = None
    if not opts.synthetic:
        # phase1/train_wgan-gp.py:24-26,71-72: MinMax-scaled still poses, a random subset sampler
        from torch.utils.data import DataLoader, SubsetRandomSampler
        from .. import data as D
        print("Loading sticks and sequences datasets...")
        dataset = D.StickDataset(runner.dataset_folder(cfg, opts.folder), normalize="minmax")
        sampler = SubsetRandomSampler(range(min(cfg["num_train"], len(dataset))))
        if device.type == "cuda" and not opts.host_loader:
            loader = D.ResidentLoader(dataset, cfg["batch_size"], sampler, device, drop_last=True)
        else:
            loader = DataLoader(dataset, batch_size=cfg["batch_size"], drop_last=True, sampler=sampler)
    logdir = runner.make_run_dir(opts.name, enabled=(rank == 0 and not opts.no_run_dir))
    np.random.seed(37)
    gen = Generator(cfg["latent_vector_size"], cfg["size"], cfg["output_size"], cfg["nblocks_gen"]).to(device)
    critic = Discriminator(cfg["output_size"], cfg["size"], cfg["nblocks_critic"]).to(device)
    engine = Phase1Engine(gen, critic, cfg, sync_bn=opts.sync_bn)
    if world > 1:  # identical weights must come from a common seed; the reference (single process) sets none
        for m in (gen, critic):
            for t in list(m.parameters()) + list(m.buffers()):
                torch.distributed.broadcast(t.data, 0)
        torch.manual_seed(torch.initial_seed() + rank)
    engine.host_noise = False  # phase1/train_wgan-gp.py:83 draws the noise on the device
    if device.type == "cuda" and (opts.graphs or (world == 1 and not opts.no_graphs)):
        engine.enable_graphs()
    log = runner.ScalarLog(logdir, opts.log_every)
    B = cfg["batch_size"]
    batches_per_epoch = max(cfg["num_train"] // B, 1)
    runner.settle_garbage_collector()
    print("Start training..")
    done = False
    for epoch in range(cfg["num_epochs"]):
        gen.train()
        def synthetic():
            for b in range(batches_per_epoch):
                g = torch.Generator().manual_seed(1 + (epoch * batches_per_epoch + b) * world + rank)
                yield torch.rand(B, 23, 3, generator=g).to(device)

        if loader is None:
            source = synthetic()
        elif isinstance(loader, DataLoader):
            source = (runner.staged((b,), device)[0][0] for b in loader)
        else:
            source = (b for b, _ in runner.resident_batches(loader, device))
        for real in source:
            out = engine.train_step(real)
            it = engine.total_iterations
            if "loss_gen" in out:
                log.scalars({"loss_critic": -out["loss_critic"], "loss_gen": out["loss_gen"]}, it)
            if opts.iterations is not None and it >= opts.iterations:
                done = True
                break
        if done:
            break
        if logdir is not None and (epoch + 1) % 5 == 0:
            engine.flush()
            runner.save_state(gen, logdir + "/models/gen_{}.pt".format(epoch + 1))
            runner.save_state(critic, logdir + "/models/critic_{}.pt".format(epoch + 1))
    engine.flush()
    log.flush()
    if rank == 0:
        print("done: {} iterations, last {}".format(engine.total_iterations,
                                                    {k: float(v) for k, v in engine.last.items()}))
    return engine


if __name__ == "__main__":
    main()
