"""Phase 2: unconditional sequence WGAN-GP/LP on MI355X.

    python -m music2dance_amd.phase2.train -c music2dance_amd/phase2/configs/default.yaml -d 0 -n run -f wgangp --synthetic

Flags -c/-d/-n/-f and YAML keys as in the reference's phase2/train.py. Only the `wgangp`
framework is part of this engine; `gan` (BCE) is outside the WGAN-GP path and, like any
unknown value, is rejected with the reference's error.
"""
import argparse

import torch

from .. import dp, runner
from ..engine import Phase2Engine
from .archis.default import SequenceDiscriminator, SequenceGenerator


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, help="choose config file")
    ap.add_argument("-d", "--device", type=int, help="choose gpu id")
    ap.add_argument("-n", "--name", type=str, help="name experiment")
    ap.add_argument("-f", "--framework", type=str, default="wgangp", help="choose between `wgangp` and `gan`")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--folder", type=str, default=None, help="dataset folder (overrides the YAML's `folder:`)")
    ap.add_argument("--iterations", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--log-every", type=int, default=1)
    ap.add_argument("--no-run-dir", action="store_true")
    ap.add_argument("--sync-bn", action="store_true", help="data parallel: BatchNorm statistics over the global batch")
    ap.add_argument("--graphs", action="store_true",
                    help="replay each loop body from captured HIP graphs (default on a single GPU: round 6 - with the "
                         "TemporalBlock kernels a loop body is 1.2 ms of GPU work and the eager loop is bound by the host's "
                         "launch rate: 16.4 k vs 19.7 k sequences / s at batch 32)")
    ap.add_argument("--no-graphs", action="store_true", help="keep the eager launch loop on a single GPU too")
    ap.add_argument("--host-loader", action="store_true",
                    help="fetch and collate batches on the host (torch DataLoader) instead of gathering them from the "
                         "HBM-resident dataset; same batches either way")
    opts = ap.parse_args(argv)
    if opts.framework != "wgangp":
        raise ValueError("Please state existing framework")

    rank, world, local = dp.init_from_env()
    device = runner.pick_device(local if world > 1 else opts.device)
    cfg = runner.load_config(opts.config)
    torch.manual_seed(0)
    ds = cfg["dataset"]
    stick_length = int(ds["seq_length"] * ds["video_rate"])
    batch_size = opts.batch_size or cfg["batch_size"]
    loader = None
    if not opts.synthetic:
        # phase2/train.py:66-70,115-116: MinMax-scaled pose sequences, random crops, a random subset sampler
        from torch.utils.data import DataLoader, SubsetRandomSampler
        from .. import data as D
        folder = runner.dataset_folder(cfg, opts.folder)
        print("Loading sticks and sequences datasets...")
        sticks = D.StickDataset(folder, normalize="minmax")
        dataset = D.SequenceDataset(folder, ds, dance_types=cfg["dance_types"], scaler=sticks.scaler, withaudio=False)
        stick_length = dataset.stick_length
        sampler = SubsetRandomSampler(range(min(cfg["num_train"], len(dataset))))
        if device.type == "cuda" and not opts.host_loader:
            loader = D.ResidentLoader(dataset, batch_size, sampler, device, drop_last=True)
        else:
            loader = DataLoader(dataset, batch_size=batch_size, drop_last=True, sampler=sampler,
                                collate_fn=lambda b: D.collate_fn(b, withaudio=False))
    logdir = runner.make_run_dir(opts.name, enabled=(rank == 0 and not opts.no_run_dir))
    gen = SequenceGenerator(cfg["input_vector_size"], cfg["latent_vector_size"], cfg["size"], cfg["output_size"],
                            cfg["nblocks_gen"], cfg["n_cells"], device)
    critic = SequenceDiscriminator(cfg["output_size"], cfg["channels"], stick_length, cfg["init_kernel"],
                                   cfg["nblocks_critic"], device)
    if next(critic.parameters()).is_cuda:
        from .. import kernels
        kernels.set_plan_model(5)   # no second critic branch to overlap with: the launch-by-launch cost model
    engine = Phase2Engine(gen, critic, cfg, sync_bn=opts.sync_bn)
    torch.manual_seed(rank)  # identical weights (seed 0 above), rank-distinct noise / alpha draws
    engine.host_noise = False  # phase2/train.py:139-140 draws the noise on the device
    if device.type == "cuda" and (opts.graphs or (world == 1 and not opts.no_graphs)):
        engine.enable_graphs()   # (data parallel: the exchange stays outside the graphs - eager there unless asked for)
    log = runner.ScalarLog(logdir, opts.log_every)
    runner.dump_architectures(logdir, gen, critic)
    batches_per_epoch = max(cfg["num_train"] // cfg["batch_size"], 1)
    runner.settle_garbage_collector()
    print("Start training..")
    done = False
    for epoch in range(cfg["num_epochs"]):
        gen.train()
        def synthetic():
            for b in range(batches_per_epoch):
                g = torch.Generator().manual_seed(1 + (epoch * batches_per_epoch + b) * world + rank)
                yield torch.rand(batch_size, stick_length, cfg["output_size"], generator=g).to(device)

        if loader is None:
            source = synthetic()
        elif isinstance(loader, DataLoader):
            source = (runner.staged((b[0].float().reshape(b[0].size(0), stick_length, -1),), device)[0][0] for b in loader)
        else:
            source = (b[0].reshape(b[0].size(0), stick_length, -1) for b, _ in runner.resident_batches(loader, device))
        for real in source:
            out = engine.train_step(real)
            it = engine.total_iterations
            if "loss_gen" in out:
                log.scalars({"loss_critic": -out["loss_critic"], "loss_gen": out["loss_gen"], "gp": out["gp"],
                             "w_dist": -out["w_dist"]}, it)
            if opts.iterations is not None and it >= opts.iterations:
                done = True
                break
        if done:
            break
        if logdir is not None and (epoch + 1) % 5000 == 0:
            engine.flush()
            runner.save_state(gen, logdir + "/models/gpgen_{}.pt".format(epoch + 1))
            runner.save_state(critic, logdir + "/models/gpcritic_{}.pt".format(epoch + 1))
    engine.flush()
    log.flush()
    if rank == 0:
        print("done: {} iterations, last {}".format(engine.total_iterations,
                                                    {k: float(v) for k, v in engine.last.items()}))
    return engine


if __name__ == "__main__":
    main()
