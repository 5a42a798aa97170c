"""Phase 3: audio-conditioned sequence WGAN-GP on MI355X.

    python -m music2dance_amd.phase3.train -c music2dance_amd/phase3/configs/default.yaml -d 0 -n run --synthetic

Same flags (-c/-d/-n), YAML keys, seeds, scalar tags and checkpoint names as the
reference's phase3/train.py; the loop body is engine.Phase3Engine.
"""
import argparse

import numpy as np
import torch

from .. import dp, runner
from ..engine import Phase3Engine, synthetic_phase3_batch
from .archis.default import AblatedSequenceDiscriminator, SequenceDiscriminator, SequenceGenerator


LAST_LOG = None  # the ScalarLog of the most recent main() (tests and notebooks read .last)


def build(cfg, device, stick_length):
    rate = cfg["dataset"]["audio_rate"]
    window = int(cfg["window_size"] * rate)
    gen = SequenceGenerator(window, cfg["input_vector_size"], cfg["latent_vector_size"], cfg["size"],
                            cfg["output_size"], cfg["noise_size"], cfg["nblocks_gen"], cfg["n_cells"],
                            cfg["enc_type"], cfg["activ"], device)
    cls = AblatedSequenceDiscriminator if cfg["ablated"] else SequenceDiscriminator
    critic = cls(cfg["output_size"], cfg["channels"], cfg["code_size"], stick_length,
                 init_ker=cfg["init_kernel"], activ=cfg["activ"], device=device)
    return gen, critic


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, help="choose config file")
    ap.add_argument("-d", "--device", type=int, help="choose gpu id")
    ap.add_argument("-n", "--name", type=str, help="name experiment")
    ap.add_argument("--synthetic", action="store_true", help="random poses / audio of the dataset's shapes")
    ap.add_argument("--folder", type=str, default=None, help="dataset folder (overrides the YAML's `folder:`)")
    ap.add_argument("--iterations", type=int, default=None, help="stop after this many loop bodies")
    ap.add_argument("--batch-size", type=int, default=None, help="override batch_size (per GPU)")
    ap.add_argument("--log-every", type=int, default=1)
    ap.add_argument("--no-run-dir", action="store_true")
    ap.add_argument("--sync-bn", action="store_true", help="data parallel: BatchNorm statistics over the global batch")
    ap.add_argument("--val-batches", type=int, default=1, help="held-out synthetic batches for l1_loss_val")
    ap.add_argument("--host-loader", action="store_true",
                    help="fetch and collate batches on the host (torch DataLoader, as the reference does) instead of "
                         "gathering them from the HBM-resident dataset; same batches either way")
    opts = ap.parse_args(argv)

    rank, world, local = dp.init_from_env()
    device = runner.pick_device(local if world > 1 else opts.device)
    cfg = runner.load_config(opts.config)
    torch.manual_seed(0)
    ds = cfg["dataset"]
    stick_length = int(ds["seq_length"] * ds["video_rate"])
    batch_size = opts.batch_size or cfg["batch_size"]
    logdir = runner.make_run_dir(opts.name, enabled=(rank == 0 and not opts.no_run_dir))
    train_loader = val_loader = None
    if not opts.synthetic:
        # phase3/train.py:72-76,112-162: scaler fitted on all still poses, sequences + audio, seeded split,
        # class-balanced samplers
        from .. import data as D
        from ..utils import slice_audio_batch
        folder = runner.dataset_folder(cfg, opts.folder)
        print("Loading sticks and sequences datasets...")
        sticks = D.StickDataset(folder, normalize="minmax")
        dataset = D.SequenceDataset(folder, ds, dance_types=cfg["dance_types"], scaler=sticks.scaler, withaudio=True)
        dataset.truncate()
        stick_length = dataset.stick_length
        resident = device if (device.type == "cuda" and not opts.host_loader) else None
        train_loader, val_loader, _ = D.make_loaders(dataset, batch_size, withaudio=True, logdir=logdir, device=resident)
        window, hop = int(cfg["window_size"] * dataset.aud_rate), dataset.ratio
    gen, critic = build(cfg, device, stick_length)
    engine = Phase3Engine(gen, critic, cfg, ablated=cfg["ablated"], sync_bn=opts.sync_bn)
    # seed 0 built identical weights on every rank; the in-loop host draws (generator noise,
    # penalty alpha) must differ between ranks, as they do between samples of one global batch
    torch.manual_seed(rank)
    global LAST_LOG
    log = LAST_LOG = runner.ScalarLog(logdir, opts.log_every)
    runner.dump_architectures(logdir, gen, critic)

    batches_per_epoch = max(cfg["num_train"] // cfg["batch_size"], 1)
    np.random.seed(14)
    n_valid_steps = 1  # phase3/train.py:168

    def loader_batches(loader):
        # the window view of the padded track is made on the copy stream, in front of the `ready` event: the
        # generator forward that reads it runs on a stream of its own and waits for nothing but that event
        def with_slices(b):
            real, audio = b[0], b[2] if len(b) > 2 else b[1]
            return real, audio, slice_audio_batch(audio, window, hop, window - hop, lazy=True)

        if isinstance(loader, D.ResidentLoader):
            for (real, audio, slices), ready in runner.resident_batches(loader, device, derive=with_slices):
                yield real, audio, slices, ready
            return
        for real_h, _, audio_h, _, _ in loader:
            (real, audio, slices), ready = runner.staged((real_h.float(), audio_h), device, derive=with_slices)
            yield real, audio, slices, ready

    def val_batches():
        # the reference's validation loader serves the held-out 20 % split as one batch
        # (phase3/train.py:161); synthetic runs: fixed held-out synthetic batches, disjoint seeds from training
        if val_loader is not None:
            for real, _, slices, _ in loader_batches(val_loader):
                yield real, slices
            return
        for v in range(opts.val_batches):
            real, _, slices = synthetic_phase3_batch(batch_size, stick_length, device, seed=-(1 + v * world + rank),
                                                     audio_rate=ds["audio_rate"], video_rate=ds["video_rate"],
                                                     window_s=cfg["window_size"])
            yield real, slices

    runner.settle_garbage_collector()
    print("Start training..")
    done = False
    e_val_loss = float("nan")
    for epoch in range(cfg["num_epochs"]):
        gen.train()
        # staged on the copy stream: the engine may start this batch's generator forward while the
        # previous iteration's critic kernels are still running
        source = loader_batches(train_loader) if train_loader is not None else (
            synthetic_phase3_batch(batch_size, stick_length, device, seed=1 + (epoch * batches_per_epoch + b) * world + rank,
                                   audio_rate=ds["audio_rate"], video_rate=ds["video_rate"],
                                   window_s=cfg["window_size"], with_event=True) for b in range(batches_per_epoch))
        for real, audio, slices, ready in source:
            out = engine.train_step(real, audio, slices, inputs_ready=ready)
            it = engine.total_iterations
            if "loss_gen" in out:
                log.scalars({"loss_critic": -out["loss_critic"], "loss_gen": out["loss_gen"], "gp": out["gp"],
                             "w_dist": -out["w_dist"], "l1_loss_train": out["l1_loss_train"]}, it)
            if opts.iterations is not None and it >= opts.iterations:
                done = True
                break
        if epoch % n_valid_steps == 0:
            # eval-mode L1 on held-out batches (phase3/train.py:245-261); validation_l1 restores train mode
            e_val = engine.validation_l1(val_batches())
            log.scalars({"l1_loss_val": e_val}, engine.total_iterations, force=True)
            last_val = e_val
        if done:
            break
        if (epoch + 1) % 500 == 0 and rank == 0:
            o = engine.last_full
            e_val_loss = float(last_val)
            print("Iteration: {} LossG : {} LossD : {} L1 train : {} L1 val : {}".format(
                engine.total_iterations, float(o.get("loss_gen", float("nan"))), float(o["loss_critic"]),
                float(o.get("l1_loss_train", float("nan"))), e_val_loss))
        if logdir is not None:
            if (epoch + 1) <= 1000 and (epoch + 1) % 100 == 0:
                runner.save_state(gen, logdir + "/models/gpgen_{}.pt".format(epoch + 1))
            if (epoch + 1) % 5000 == 0:
                engine.flush()  # a deferred (data-parallel) critic step must be in the checkpoint
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
