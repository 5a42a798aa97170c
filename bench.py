#!/usr/bin/env python
"""Throughput of the phase-3 audio-conditioned WGAN-GP training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one loop body of phase3/train.py:186-243 on one synthetic batch already
resident in HBM: a critic iteration (generator forward, gradient penalty with its double
backward, real/fake critic passes, critic backward, Adam), plus a generator iteration on
every n_critic_steps-th step (8, phase3/configs/default.yaml) — so K should be a multiple
of 8 to time whole cycles. Workload = BASELINE.json configs[2]: default conv1d encoder,
120-frame sequences, batch 64 per GPU (weak scaling: the global batch is 64 * N).

Rank 0 prints ONE JSON line: seq/s over all GPUs, plus
  roofline     — the implicit-GEMM engine (conv1d fwd/bwd_data/bwd_weight + linear GEMMs):
                 sum of 2*M*N*K over its launches / sum of their HIP-event durations,
                 measured inside the timed region on the launch stream, against the exact
                 fp32 MFMA peak of gfx950 (157.3 TFLOP/s);
  cpu_baseline — the oracle (a port of the reference's CPU path, oracle/m2d_oracle.py) timed
                 on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# GFLOP the reference's own formulation executes per sequence consumed per 8+1 cycle, by workload
# (encoder, frames, ablated critic); shapes outside the table report no equivalent-work rate
REF_GFLOP_PER_SEQ_CYCLE = {("default", 120, False): 31.10, ("wavegan", 120, False): 35.91,
                           ("unet", 120, False): 52.12, ("unet", 300, True): 65.3,
                           "phase2": 2.772, "phase1": 0.864e-3}

P3_DEFAULT = {"lr_gen": 2e-4, "lr_critic": 2e-4, "n_critic_steps": 8, "gamma": 10, "beta": 1, "eta": 0,
              "output_size": 69}


def build_models(device, seqlen=120, enc_type="default", ablated=False):
    from music2dance_amd.phase3.archis.default import (AblatedSequenceDiscriminator, SequenceDiscriminator,
                                                       SequenceGenerator)
    torch.manual_seed(0)  # phase3/train.py:35
    gen = SequenceGenerator(3200, 250, 250, 256, 69, 10, 2, 3, enc_type, "id", device)
    cls = AblatedSequenceDiscriminator if ablated else SequenceDiscriminator
    critic = cls(69, 128, 100, seqlen, init_ker=25, activ="id", device=device)
    return gen, critic


def pmc_traffic():
    """HBM bytes per engine launch from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs of this same command, profiles/*_pmc_traffic.json);
    PMC counters cannot be read from inside the process, so this is the recorded value."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    # gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B requests at 64 B -> x2
    return d.get("hbm_bytes_per_launch_fetch_x2", d.get("hbm_bytes_per_launch_uncorrected")), os.path.basename(files[-1])


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(full_b=64, sample_b=32, T=120, threads=None, cap_s=75.0):
    """The oracle (CPU port of the reference path, oracle/m2d_oracle.py) timed on this box's host cores.
    A MEASUREMENT when it fits the time cap: one warm-up critic iteration at the full batch 64, then ONE whole 8+1
    cycle (8 critic iterations + the generator iteration, phase3/train.py:186-243) timed end to end -> 8 * 64 sequences
    / cycle time. If the warm-up says the cycle would exceed `cap_s` seconds (the default bench run has
    to finish within minutes) it falls back to the round-3 sample: batch 32, 2 critic + 2 (critic + generator)
    iterations, extrapolated to a cycle (the path is linear in the batch) - and says so in `sample`.
    Thread count: the op sizes here stop scaling beyond ~32 intra-op threads on the GPU box's host (measured at batch 8
    per critic iteration: 16 threads 0.51 s, 32 threads 0.45 s, 64 threads 1.5 s, 128 threads 4.1 s), so the baseline uses
    min(32, visible cores) and reports that number as `cores`."""
    from oracle import m2d_oracle as O
    from music2dance_amd.engine import synthetic_phase3_batch
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = threads or min(32, visible)
    torch.set_num_threads(cores)
    gen, critic = build_models("cpu")
    gsd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    dsd = {k: v.detach().clone() for k, v in critic.state_dict().items()}
    host = {"cpu_model": _cpu_model(), "visible_cores": visible, "torch": torch.__version__,
            "thread_sweep": "batch 8 critic iteration: 16 thr 0.51 s, 32 thr 0.45 s, 64 thr 1.5 s, 128 thr 4.1 s (round 2, this pool)"}
    crit_only = O.P3Config(n_critic=10 ** 9)
    real, audio, slices = synthetic_phase3_batch(full_b, T, "cpu", seed=1)
    t0 = time.perf_counter()
    O.p3_train_iterations(gsd, dsd, crit_only, real, audio, slices, 1, 0)  # warm-up critic iteration at the full batch
    t_w = time.perf_counter() - t0
    if 9.5 * t_w <= cap_s:
        t1 = time.perf_counter()
        trace, _, _ = O.p3_train_iterations(gsd, dsd, O.P3Config(n_critic=8), real, audio, slices, 8, PARITY_SEED)  # 8 critic + 1 generator
        cycle = time.perf_counter() - t1
        # the cycle just timed is also a B = 64 loss trace from known weights / batch / draws: parity_at_bench_size()
        # replays it on the GPU (outside every timed region)
        host["_parity"] = dict(gsd=gsd, dsd=dsd, batch=(real, audio, slices), n_critic=8, iters=8, trace=trace)
        return dict({"value": round(8.0 * full_b / cycle, 3), "unit": "seq/s", "cores": cores, "kind": "port",
                     "sample": "MEASURED: one whole 8+1 cycle of the oracle's phase-3 default-encoder step at batch %d "
                               "(%.1f s, after one warm-up critic iteration of %.1f s)" % (full_b, cycle, t_w)}, **host)
    real, audio, slices = synthetic_phase3_batch(sample_b, T, "cpu", seed=1)
    t0 = time.perf_counter()
    O.p3_train_iterations(gsd, dsd, crit_only, real, audio, slices, 1, 0)  # warm-up critic iteration
    t1 = time.perf_counter()
    O.p3_train_iterations(gsd, dsd, crit_only, real, audio, slices, 2, 0)  # 2 critic iterations
    t2 = time.perf_counter()
    trace, _, _ = O.p3_train_iterations(gsd, dsd, O.P3Config(n_critic=1), real, audio, slices, 2, PARITY_SEED)  # 2 x (critic + generator)
    t3 = time.perf_counter()
    host["_parity"] = dict(gsd=gsd, dsd=dsd, batch=(real, audio, slices), n_critic=1, iters=2, trace=trace)
    t_critic = (t2 - t1) / 2.0
    t_gen = max((t3 - t2) / 2.0 - t_critic, 0.0)
    cycle = 8 * t_critic + t_gen
    return dict({"value": round(8.0 * sample_b / cycle, 3), "unit": "seq/s", "cores": cores, "kind": "port",
                 "sample": "EXTRAPOLATED (a batch-%d critic iteration took %.1f s: a whole cycle would exceed the %d s cap): "
                           "oracle step at batch %d, 5 critic + 2 generator iterations (%.1f s), scaled to one 8+1 cycle; "
                           "critic %.2f s, generator %.2f s per iteration"
                           % (full_b, t_w, int(cap_s), sample_b, t3 - t0, t_critic, t_gen)}, **host)


PARITY_SEED = 0  # host-generator seed of the oracle's timed cycle and of its GPU replay (noise / alpha draws)


def parity_at_bench_size(pay, device):
    """The oracle cycle `cpu_baseline` has just timed, replayed by a FRESH engine on the GPU: same constructor-seeded
    weights, same synthetic batch (drawn on the host, copied over), same host-generator seed, so the generator noise and
    the penalty's interpolation weights are the same draws in the reference's order (phase3/archis/default.py:31-34,
    losses.py:15). Eager, no pipelining (no `inputs_ready`): the draw order is the reference's. Runs after the timed
    region and after the roofline pass; the oracle is the checker here, never the thing measured.
    -> {"steps", "batch", per scalar: worst relative error over the steps, "worst_rel"}"""
    from music2dance_amd.engine import Phase3Engine
    gen, critic = build_models("cpu")
    gen.load_state_dict(pay["gsd"]), critic.load_state_dict(pay["dsd"])
    gen.to(device), critic.to(device)
    gen.train(), critic.train()
    cfg = dict(P3_DEFAULT, n_critic_steps=pay["n_critic"])
    eng = Phase3Engine(gen, critic, cfg, data_parallel=False)
    from music2dance_amd.utils import slice_audio_batch
    real, audio = (t.to(device) for t in pay["batch"][:2])
    win = pay["batch"][2].shape[-1]
    hop = audio.shape[1] // pay["batch"][2].shape[1]
    slices = slice_audio_batch(audio, win, hop, win - hop, lazy=True)  # the window view the timed region uses
    torch.manual_seed(PARITY_SEED)
    got = {"loss_critic": [], "gp": [], "w_dist": [], "loss_gen": [], "l1_loss_train": []}
    for _ in range(pay["iters"]):
        out = eng.train_step(real, audio, slices)
        for k in got:
            if k in out:
                got[k].append(float(out[k]))
    eng.flush()
    tr = pay["trace"]
    want = {"loss_critic": tr["loss_critic"], "gp": tr["gp"], "w_dist": tr["w_dist"], "loss_gen": tr["loss_gen"],
            "l1_loss_train": tr["err_l1"]}
    res = {"steps": pay["iters"], "batch": int(real.shape[0]), "generator_iterations": len(want["loss_gen"]),
           "rel": "|gpu - oracle| / max(|oracle|, 1e-3), worst over the steps"}
    worst = 0.0
    for k, w in want.items():
        g = got[k]
        if len(g) != len(w):
            res[k] = "length mismatch: %d vs %d" % (len(g), len(w))
            worst = float("inf")
            continue
        e = max([abs(a - b) / max(abs(b), 1e-3) for a, b in zip(g, w)] + [0.0])
        res[k] = float("%.3e" % e)
        worst = max(worst, e)
    res["worst_rel"] = float("%.3e" % worst)
    res["oracle_last"] = {k: (w[-1] if w else None) for k, w in want.items()}
    res["gpu_last"] = {k: (g[-1] if g else None) for k, g in got.items()}
    return res


def dist_info(rank, world, local, device, engine, waits):
    """What a reader of a multi-GPU line needs to check that it ran as claimed: world size and backend as
    torch.distributed reports them, the RCCL version, every rank's device, and how many gradient buckets left
    underneath the backward pass (dp.GradExchange.launched_in_backward) on this rank."""
    info = {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "backend": dist.get_backend() if dist.is_initialized() else None,
            "rccl_version": None, "devices": None}
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        pass
    mine = "rank %d: cuda:%d %s" % (rank, local, torch.cuda.get_device_name(device))
    if dist.is_initialized() and world > 1:
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        info["devices"] = allr
    else:
        info["devices"] = [mine]
    for name in ("x_critic", "x_gen"):
        x = getattr(engine, name, None)
        if x is not None:
            info[name] = {"active": bool(x.active), "buckets": len(x.buckets),
                          "launched_in_backward": int(x.launched_in_backward)}
            # time the consuming stream sat in GradExchange.finish() per exchange of the timed steps (this rank): the
            # part of the all-reduce that did not hide under the backward pass / the next generator forward
            w = waits[name]
            info[name].update({"exchanges_timed": w["exchanges"], "exchange_wait_ms": w["wait_ms_mean"],
                               "exchange_wait_ms_max": w["wait_ms_max"]})
    # time the consumer of the pipelined generator forward waited for it (per join of the timed steps, this rank)
    j = waits.get("join")
    if j is not None:
        info["gen_forward_joins_timed"] = j["joins"]
        info["gen_forward_join_ms"] = j["wait_ms_mean"]
        info["gen_forward_join_ms_max"] = j["wait_ms_max"]
    if dist.is_initialized() and world > 1:
        mine2 = {"rank": rank, "exchange_wait_ms": {n: info[n]["exchange_wait_ms"] for n in ("x_critic", "x_gen") if n in info},
                 "gen_forward_join_ms": info.get("gen_forward_join_ms")}
        allr = [None] * world
        dist.all_gather_object(allr, mine2)
        info["per_rank_waits"] = allr
    return info


def arm_watchdog():
    """A rank that hangs (a collective whose peer died, a persistent kernel that never returns) must not hold the
    launcher for ever: after M2D_BENCH_TIMEOUT seconds (default 1800) the process dumps its Python stacks and exits
    non-zero from a timer thread - no re-exec, no signal to anybody else."""
    import faulthandler
    import threading
    limit = float(os.environ.get("M2D_BENCH_TIMEOUT", "1800"))

    def fire():
        sys.stderr.write("bench.py: rank %s exceeded %.0f s - exiting\n" % (os.environ.get("RANK", "0"), limit))
        faulthandler.dump_traceback(file=sys.stderr)
        sys.stderr.flush()
        os._exit(4)

    t = threading.Timer(limit, fire)
    t.daemon = True
    t.start()
    return t


def parity_c2(device, B=32, T=120, iters=3):
    """BASELINE configs[1] at its bench size against the oracle: `iters` critic iterations of the phase-2 loop
    (phase2/train.py:135-156) at the config's own lr 5e-4 from constructor-seeded weights, the same host draws (noise, then
    alpha, per iteration). The closed-form LP critic amplifies fp32 rounding step by step (tests/test_product_parity.py:
    test_p2_trace_at_the_config_learning_rate binds the reference-generated trace at 1e-4 / 2e-3 / 1e-2 for steps 1-3);
    the same bounds are applied here. The oracle is the checker; nothing here is timed."""
    from oracle import m2d_oracle as O
    from music2dance_amd.engine import Phase2Engine
    from music2dance_amd.phase2.archis.default import SequenceDiscriminator as D2, SequenceGenerator as G2
    torch.manual_seed(0)
    gen = G2(50, 50, 256, 69, 2, 3, "cpu")
    critic = D2(69, 128, T, 25, 3, "cpu")
    gsd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    dsd = {k: v.detach().clone() for k, v in critic.state_dict().items()}
    real = torch.rand(B, T, 69, generator=torch.Generator().manual_seed(7))
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8))
    t0 = time.perf_counter()
    want, _, _ = O.p2_train_iterations(gsd, dsd, real, iters, PARITY_SEED, lr=P2_DEFAULT["lr_critic"], n_critic=10 ** 9, T=T)
    t_oracle = time.perf_counter() - t0
    gen.to(device), critic.to(device)
    gen.train(), critic.train()
    eng = Phase2Engine(gen, critic, dict(P2_DEFAULT, n_critic_steps=10 ** 9), data_parallel=False)
    eng.host_noise = True   # the oracle draws on the host generator
    torch.manual_seed(PARITY_SEED)
    got = {"loss_critic": [], "gp": [], "w_dist": []}
    realg = real.to(device)
    for _ in range(iters):
        out = eng.train_step(realg)
        for k in got:
            got[k].append(float(out[k]))
    eng.flush()
    bounds = (1e-4, 2e-3, 1e-2)
    res = {"steps": iters, "batch": B, "oracle_s": round(t_oracle, 2), "bounds_per_step": list(bounds[:iters]),
           "rel": "|gpu - oracle| / max(|oracle|, 1), per step"}
    ok = True
    for k in got:
        errs = [abs(a - b) / max(abs(b), 1.0) for a, b in zip(got[k], want[k])]
        res[k] = [float("%.3e" % e) for e in errs]
        ok = ok and all(e <= bounds[min(i, len(bounds) - 1)] for i, e in enumerate(errs))
    res["within_bounds"] = bool(ok)
    return res


def other_configs(steps=16, warmup=8):
    """The other BASELINE presets at their per-GPU shapes, each as a CHILD process, one after the other, before this
    process touches the GPU: what the driver's one default command would otherwise never see (round-5 verdict item 4). -> {preset: {value, ms_per_step, whole cycle figures, engine / whole-step / pose-critic fractions, memory-bound
    families, parity where the oracle fits the cap}}. A failing child is reported, never fatal."""
    import subprocess
    out = {}
    for name in ("c2", "c4", "c5"):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(steps), "--warmup", str(warmup),
               "--no-cpu-baseline", "--no-other-configs"] + (["--parity-check"] if name == "c2" else [])
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get("M2D_OTHER_TIMEOUT", "240")))
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"failed": "rc %d: %s" % (r.returncode, (r.stderr or "")[-300:])}
                continue
            d = json.loads(line[-1])
        except Exception as e:  # informational: never lose the headline over it
            out[name] = {"failed": repr(e)}
            continue
        rf = d.get("roofline", {})
        e = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
             "steps": d["steps"], "warmup": d["warmup"], "wall_s": round(time.perf_counter() - t0, 1)}
        if "whole_cycles" in d:
            e["whole_cycle_value"] = d["whole_cycles"]["value"]
            e["whole_cycle_ms_per_step"] = d["whole_cycles"]["ms_per_step"]
        e["engine_frac"] = rf.get("frac")
        e["whole_step_frac"] = rf.get("whole_step_frac")
        if "tcn_critic" in rf:
            e["tcn_critic_frac"] = rf["tcn_critic"]["frac"]
            e["tcn_critic_ms_per_step"] = rf["tcn_critic"]["ms_per_step"]
        e["kernel_ms_per_step"] = rf.get("kernel_ms_per_step")
        e["hbm_families_frac"] = {k: v["frac"] for k, v in rf.get("hbm", {}).items() if "frac" in v}
        if "parity_at_bench_size" in d:
            e["parity_at_bench_size"] = d["parity_at_bench_size"]
        e["async_faults"] = d.get("async_faults")
        out[name] = e
    return out


def step_clock():
    """The shader clock inside the step's largest launches, from the stamped build of the library (libm2d_hip_stamp.so:
    per-workgroup s_memrealtime / s_memtime stamps around the K loop) - a child process (tools/step_clock.py) run before
    this process touches the GPU. -> its JSON ({"shapes": [...], "clock_GHz_in_step"}) or {"failed": ...}."""
    import subprocess
    from music2dance_amd import build as m2d_build
    lib = m2d_build.STAMP_LIB_PATH
    if not os.path.exists(lib):
        return {"failed": "no stamped library (python -m music2dance_amd.build)"}
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_clock.py")], capture_output=True, text=True,
                           env=dict(os.environ, M2D_LIB=lib), timeout=180)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"failed": "rc %d: %s" % (r.returncode, (r.stderr or "")[-300:])}
        return json.loads(line[-1])
    except Exception as e:
        return {"failed": repr(e)}


PRESETS = {  # BASELINE.json configs[0..4] at their per-GPU shapes
    # phase1/configs/b1l10s128.yaml (BASELINE words it "CPU": GPU-only here). Host-launch-bound when eager (2.8-3.2 ms
    # per body for 0.75 ms of kernels): the preset replays the captured graphs (Phase1Engine.enable_graphs, 0.86 ms)
    "c1": dict(phase=1, batch=64, frames=1, graphs="on"),
    # phase2/configs/default.yaml at batch 32. Round 6: with the TemporalBlock kernels a loop body is 1.2 ms of GPU work
    # and the eager loop is bound by the host's launch rate (16.4 k vs 19.7 k seq/s, three alternating pairs on one box):
    # the preset replays the captured graphs, as the phase-2 train script now does on one GPU (--graphs off: eager)
    "c2": dict(phase=2, batch=32, frames=120, graphs="on"),
    "c3": dict(enc_type="default", batch=64, frames=120, ablated=False),
    "c4": dict(enc_type="wavegan", batch=32, frames=120, ablated=False),  # global 256 on 8 GPUs
    "c5": dict(enc_type="unet", batch=16, frames=300, ablated=True),     # global 128 on 8 GPUs
}
P2_DEFAULT = {"lr_gen": 5e-4, "lr_critic": 5e-4, "n_critic_steps": 8, "gamma": 10, "eta": 50, "input_vector_size": 50,
              "output_size": 69}
P1_DEFAULT = {"lr_gen": 1e-4, "lr_critic": 1e-4, "n_critic_steps": 5, "gamma": 10, "latent_vector_size": 10}


def baseline_tag(args):
    if args.phase != 3:
        return ""
    key = dict(enc_type=args.enc_type, batch=args.batch, frames=args.frames, ablated=args.ablated)
    for name, idx in (("c3", 2), ("c4", 3), ("c5", 4)):
        if PRESETS[name] == key:
            return " (BASELINE.json configs[%d]%s)" % (idx, "" if idx == 2 else " per-GPU shape")
    return ""


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as fresh child processes
    (this parent has made no GPU call), relay rank 0's JSON line, fail if any rank fails."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread (its pipe must not fill up while we poll); every child is polled: the
    # first failing rank ends the others (they would otherwise sit in a rendezvous / all-reduce forever), and the
    # whole run is bounded (M2D_BENCH_TIMEOUT seconds, default 3600)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("M2D_BENCH_TIMEOUT", "3600"))
    failed = None
    while failed is None:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = "ranks failed (rank, exit code): %s" % bad
        elif time.monotonic() > deadline:
            failed = "timed out; unfinished ranks: %s" % [r for r, c in enumerate(codes) if c is None]
        else:
            time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    bad = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
    if failed is not None or bad:
        sys.exit("bench.py: %s" % (failed or "ranks failed (rank, exit code): %s" % bad))


PEAK_HBM_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FAMILIES = ["gemm", "bn", "gru", "pointwise", "reduce"]


def hbm_table(launches, steps):
    """launches: kernels.prof_dump() rows. -> ({tag: {...}} for the memory-bound families, per-shape rows)."""
    import collections
    agg = collections.OrderedDict()
    for fam, tag, d0, d1, d2, ms, flops, nbytes in launches:
        a = agg.setdefault((fam, tag, d0, d1, d2), [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms
        a[2] += flops
        a[3] += nbytes
    shapes = []
    for (fam, tag, d0, d1, d2), (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        shapes.append((FAMILIES[fam], tag, d0, d1, d2, round(n / steps, 2), round(ms / steps, 4), round(1e3 * ms / n, 2),
                       round(fl / ms / 1e9, 2) if ms > 0 else 0.0, round(by / ms / 1e6, 1) if ms > 0 else 0.0))
    hbm = {}
    for fam, tag, d0, d1, d2, ms, flops, nbytes in launches:
        if FAMILIES[fam] == "gemm" or FAMILIES[fam] == "gru" or nbytes <= 0:
            continue
        big = nbytes >= 64e6
        h = hbm.setdefault(tag or FAMILIES[fam], {"large": [0, 0.0, 0.0], "small": [0, 0.0, 0.0]})
        b = h["large" if big else "small"]
        b[0] += 1
        b[1] += ms
        b[2] += nbytes
    out = {}
    for tag, h in hbm.items():
        e = {}
        n, ms, by = h["large"]
        if n:
            gbps = by / ms / 1e6
            e.update({"launches_per_step": round(n / steps, 2), "MB_per_launch": round(by / n / 1e6, 1),
                      "us_per_launch": round(1e3 * ms / n, 1), "GBps": round(gbps, 0),
                      "frac": round(gbps / PEAK_HBM_GBPS, 3)})
        n, ms, by = h["small"]
        if n:
            e["small"] = {"launches_per_step": round(n / steps, 2), "us_per_launch": round(1e3 * ms / n, 1),
                          "MB_per_launch": round(by / n / 1e6, 2)}
        out[tag] = e
    return out, shapes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None, choices=sorted(PRESETS),
                    help="BASELINE.json config preset at its per-GPU shape: c3 (default), c4 wavegan B=32, "
                         "c5 unet T=300 ablated B=16; explicit flags below override nothing when given")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=64, help="sequences per GPU (weak scaling)")
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--enc-type", default="default", choices=["default", "unet", "wavegan"],
                    help="audio encoder (BASELINE.json configs[3]/[4] use wavegan / unet)")
    ap.add_argument("--ablated", action="store_true", help="pose-only critic (required for --frames != 120)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for single-GPU dry runs)")
    ap.add_argument("--same-device", action="store_true", help="debug: all ranks on cuda:0 (with --backend gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the HIP-event kernel profile")
    ap.add_argument("--dump-shapes", default=None, metavar="CSV",
                    help="write the roofline pass's per-shape table (family, tag, dims, launches, ms, TF/s or GB/s)")
    ap.add_argument("--graphs", default=None, choices=["on", "off"],
                    help="replay each loop body's forward/backward from a captured HIP graph (Phase3Engine.enable_graphs)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default workload only: do not run the c2 / c4 / c5 presets (child processes, after every timed "
                         "region of this one) for the `other_configs` entry of the line")
    ap.add_argument("--parity-check", action="store_true",
                    help="presets that have one (c2): replay a short oracle trace at the bench size on the GPU afterwards")
    ap.add_argument("--phase", type=int, default=3, choices=[1, 2, 3],
                    help="which train script's loop body (3 = the BASELINE metric; 1 / 2: --config c1 / c2)")
    args = ap.parse_args()
    if args.config:
        for k, v in PRESETS[args.config].items():
            if k == "graphs" and args.graphs is not None:
                continue   # (an explicit --graphs on / off wins over the preset's)
            setattr(args, k, v)
    if args.graphs is None:
        args.graphs = "off"
    if args.same_device:
        os.environ["M2D_PERSISTENT_GRU"] = "0"  # ranks sharing one GPU: persistent kernels could starve each other
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)  # before anything touches the GPU
    # The other BASELINE presets (c2 / c4 / c5) for the `other_configs` entry: child processes, run to completion BEFORE
    # this process makes its first GPU call - each has the device to itself (measured the other way round, with this
    # process's idle HIP context alive beside the child, the launch-bound c2 loop ran 13 % slower: 15.2 k vs 17.5 k
    # seq/s) - and none of it overlaps this workload's timed region.
    others = clock = None
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_other_configs and args.phase == 3 and
            (args.enc_type, args.frames, args.ablated, args.batch) == ("default", 120, False, 64)):
        others = other_configs()
        clock = step_clock()

    from music2dance_amd import dp, kernels, runner
    from music2dance_amd.engine import Phase3Engine, synthetic_phase3_batch

    watchdog = arm_watchdog()
    rank, world, local = dp.init_from_env(args.backend)
    if args.same_device:
        local = 0
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: the product has no CPU path")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    # launch-plan cost model (kernels.set_plan_model): 5 where the loop body has no second critic branch to overlap its
    # launches with (phase 2, the pose-only critic of configs[4]); set once, before the first launch of the process
    if args.phase == 2 or (args.phase == 3 and args.ablated):
        kernels.set_plan_model(5)
    if args.phase == 3:
        if args.frames != 120 and not args.ablated:
            sys.exit("the audio critic only accepts 76 800-sample (120-frame) audio: use --ablated with --frames %d"
                     % args.frames)
        gen, critic = build_models(device, args.frames, args.enc_type, args.ablated)
        engine = Phase3Engine(gen, critic, P3_DEFAULT, ablated=args.ablated)
        real, audio, slices = synthetic_phase3_batch(args.batch, args.frames, device, seed=100 + rank)
        batch = (real, audio, slices)
        workload = ("phase3/train.py WGAN-GP step, %s audio encoder%s, %d frames, batch %d per GPU%s; 1 generator "
                    "iteration per 8 critic iterations" % (args.enc_type, ", ablated critic" if args.ablated else "",
                                                           args.frames, args.batch, baseline_tag(args)))
        ref_key = (args.enc_type, args.frames, args.ablated)
    elif args.phase == 2:
        from music2dance_amd.engine import Phase2Engine
        from music2dance_amd.phase2.archis.default import SequenceDiscriminator as D2, SequenceGenerator as G2
        torch.manual_seed(0)
        gen = G2(50, 50, 256, 69, 2, 3, device)
        critic = D2(69, 128, args.frames, 25, 3, device)
        engine = Phase2Engine(gen, critic, P2_DEFAULT)
        engine.host_noise = False  # phase2/train.py:139-140 draws the noise on the device
        batch = (torch.rand(args.batch, args.frames, 69, generator=torch.Generator().manual_seed(100 + rank)).to(device),)
        workload = ("phase2/train.py WGAN-LP step (GRU generator, 3-block TCN critic), %d frames, batch %d per GPU%s; "
                    "1 generator iteration per 8 critic iterations"
                    % (args.frames, args.batch, " (BASELINE.json configs[1])" if (args.frames, args.batch) == (120, 32) else ""))
        ref_key = "phase2"
    else:
        from music2dance_amd.engine import Phase1Engine
        from music2dance_amd.phase1.archis.residual import Discriminator as D1, Generator as G1
        torch.manual_seed(0)
        gen, critic = G1(10, 128, 69, 1).to(device), D1(69, 128, 1).to(device)
        engine = Phase1Engine(gen, critic, P1_DEFAULT)
        engine.host_noise = False  # phase1/train_wgan-gp.py:83 draws the noise on the device
        batch = (torch.rand(args.batch, 23, 3, generator=torch.Generator().manual_seed(100 + rank)).to(device),)
        workload = ("phase1/train_wgan-gp.py still-pose WGAN-GP step (residual MLPs, 1 block, width 128), batch %d per "
                    "GPU%s; 1 generator iteration per 5 critic iterations"
                    % (args.batch, " (BASELINE.json configs[0]; on the GPU: the product has no CPU path)" if args.batch == 64 else ""))
        ref_key = "phase1"
    if args.graphs == "on":
        if not hasattr(engine, "enable_graphs"):
            sys.exit("--graphs on: this phase's engine has no captured-graph mode")
        engine.enable_graphs()
    gen.train(), critic.train()
    # the synthetic batch is resident and complete from here on: lets the engine start an iteration's
    # generator forward on its second stream while the previous iteration's critic kernels still run
    torch.cuda.synchronize(device)
    ready = torch.cuda.current_stream(device).record_event()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # A generator iteration runs on every 8th loop body. Whatever --steps / --warmup are, the timed window starts at
    # a phase of that cycle where it holds ceil(steps / 8) generator iterations (never fewer than the nominal
    # steps / 8: the number cannot flatter), and the roofline pass below restarts at the same phase, so both passes
    # execute the same work. `whole_cycles` in the output is the phase-independent figure: the mean GPU time of
    # every window of 8 consecutive timed steps (each holds exactly one generator iteration).
    ncs = engine.n_critic_steps
    want_gen = -(-args.steps // ncs)
    phase = next(o for o in range(ncs) if sum(1 for i in range(o + 1, o + args.steps + 1) if i % ncs == 0) == want_gen)
    torch.manual_seed(1234 + rank)
    engine.total_iterations = (phase - args.warmup) % ncs  # the warm-up ends exactly at `phase`
    for _ in range(args.warmup):
        engine.train_step(*batch, inputs_ready=ready)
    engine.flush()
    assert engine.total_iterations % ncs == phase
    # host hygiene, as the train scripts do after building their engine: a generation-2 garbage collection over
    # the module graph stalls the launch thread for 60-80 ms (tools/spike_probe.py)
    runner.settle_garbage_collector()
    K = kernels.impl()
    faults0 = int(getattr(type(K), "async_faults", 0))
    barrier()
    for x in (getattr(engine, "x_critic", None), getattr(engine, "x_gen", None)):
        if x is not None:
            x.wait_stats(reset=True)   # the wait diagnostics cover the timed steps only
    if hasattr(engine, "join_stats"):
        engine.join_stats(reset=True)
    # one event per step on the main stream (asynchronous: no host sync) for the per-cycle figure
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    step_events[0].record()
    for i in range(args.steps):
        engine.train_step(*batch, inputs_ready=ready)
        step_events[i + 1].record()
    engine.flush()
    barrier()
    elapsed = time.perf_counter() - t0
    # persistent-kernel timeouts recovered from inside the timed window (engine._check_async clears the word as it
    # recovers, so K.check_async_errors() below cannot see them): their optimizer steps were voided on the device -
    # a rate over such a window is not a training rate. Reported on the line; the run then exits non-zero.
    faults_timed = int(getattr(type(K), "async_faults", 0)) - faults0
    # the wait diagnostics of the timed steps, read now (the device is idle; the roofline pass below adds its own)
    waits = {n: getattr(engine, n).wait_stats() for n in ("x_critic", "x_gen") if getattr(engine, n, None) is not None}
    waits["join"] = engine.join_stats() if hasattr(engine, "join_stats") else None
    step_ms = [a.elapsed_time(b) for a, b in zip(step_events, step_events[1:])]
    if os.environ.get("M2D_STEP_TIMES"):  # dev aid: per-step GPU time to stderr
        print("step ms:", " ".join("%.2f" % v for v in step_ms), file=sys.stderr)
    whole_cycles = None
    if args.steps >= ncs:
        wins = [sum(step_ms[i:i + ncs]) for i in range(args.steps - ncs + 1)]
        ms_cycle = sum(wins) / len(wins)
        whole_cycles = {"ms_per_cycle": round(ms_cycle, 3), "ms_per_step": round(ms_cycle / ncs, 3),
                        "value": round(ncs * args.batch * world / (ms_cycle * 1e-3), 2),
                        "windows": len(wins), "generator_iterations_in_timed_steps": want_gen,
                        "note": "mean GPU time (rank 0's stream events) of every window of %d consecutive timed steps "
                                "= exactly one generator iteration each; `value` above is the contract's K steps / "
                                "wall time, whose window holds ceil(K/%d) generator iterations" % (ncs, ncs)}
    # Per-launch HIP events for the roofline: a SECOND pass over the same K steps. Two event
    # records around each of the ~680 launches of a step cost ~8 % of wall time (measured:
    # 20.3 vs 18.7 ms per step), so they stay out of the region `value` is timed on; the kernels,
    # their order and their inputs are the same.
    prof = None
    if not args.no_prof:
        if args.graphs == "on":
            engine.enable_graphs(False)  # graph replays carry no per-launch events
        # the timed region runs the critic's pose branch on a side stream under the audio branch; an
        # event pair around a launch would then also count the other stream's kernels, so the roofline
        # pass runs the two branches one after the other: each duration is the launch's own
        if hasattr(type(critic), "overlap_branches"):
            type(critic).overlap_branches = False
        if engine.manual_critic is not None:
            engine.manual_critic.overlap = False
        engine.pipeline_generator = False  # likewise the generator forward: in line for the per-launch pass
        engine.total_iterations = phase  # same place in the 8+1 cycle as the timed window: same work
        K.prof_begin()
        for _ in range(args.steps):
            engine.train_step(*batch)
        engine.flush()
        barrier()
        launches = K.prof_dump()
        prof = K.prof_end()

    K.check_async_errors()
    t = torch.tensor([elapsed, float(faults_timed)], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, faults_timed = t[0].item(), int(t[1].item())
    last = {k: float(v) for k, v in engine.last.items()}
    dinfo = dist_info(rank, world, local, device, engine, waits)

    if rank == 0:
        seqs = args.steps * args.batch * world
        value = seqs / elapsed
        out = {
            "metric": ("%d-frame seq/sec, phase%d WGAN-GP step, batch %d per GPU" % (args.frames, args.phase, args.batch))
            if args.phase != 1 else "poses/sec, phase1 WGAN-GP step, batch %d per GPU" % args.batch,
            "value": round(value, 2), "unit": "seq/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "global_batch": args.batch * world, "seq_len": args.frames,
                       "parallelism": "dp%d" % world, "backend": args.backend if world > 1 else None},
            "losses_last_step": last,
            # recovered persistent-kernel timeouts inside the timed window, max over ranks (0 = every step was taken)
            "async_faults": faults_timed,
            "distributed": dinfo,
        }
        if whole_cycles is not None:
            out["whole_cycles"] = whole_cycles
        if prof is not None:
            g = prof["gemm"]
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
            # (the committed PMC passes are of the default workload: other presets report no traffic)
            c3_default = args.phase == 3 and (args.enc_type, args.frames, args.ablated, args.batch) == ("default", 120, False, 64)
            traffic, traffic_src = pmc_traffic() if c3_default else (None, None)
            hbm, shapes = hbm_table(launches, args.steps)
            if args.dump_shapes:
                with open(args.dump_shapes, "w") as f:
                    f.write("family,tag,d0,d1,d2,launches_per_step,ms_per_step,us_per_launch,tflops,gbps\n")
                    for row in shapes:
                        f.write(",".join(str(v) for v in row) + "\n")
            out["roofline"] = {
                "measured_over": "second pass of the same %d steps with HIP events around every launch, "
                                 "all work on one stream (in the timed region the critic's pose branch runs on a side "
                                 "stream under its audio branch, and the critic iterations' generator forward on a "
                                 "second stream under the previous iteration's critic kernels)" % args.steps,
                "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "m2d_gemm_kernel (separable-gather fp32 MFMA engine; conv1d fwd/bwd_data/bwd_weight + linear)",
                "launches_per_step": round(g["launches"] / args.steps, 1),
                "avg_launch_us": round(1e3 * g["ms"] / max(g["launches"], 1), 2),
                "gflop_per_step_executed": round(g["flops"] / args.steps / 1e9, 1),
                # every engine FLOP of a step over the step's wall time (all other kernels and gaps included)
                "whole_step_frac": round(g["flops"] / elapsed / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items()},
                # memory-bound kernel families: algorithmic bytes / event time against 8.0 TB/s, over the
                # launches that move >= 64 MB (smaller ones are launch-bound, listed under "small")
                "hbm": hbm,
            }
            # What this part's memory system delivers to the simplest possible kernels at the size of the memory-bound
            # families' tensors (157 MB = the audio critic's first activation at B = 64), measured here and now: a fill
            # (write only), a copy (read + write) and a sum (read only), 50 launches each behind one event pair. The
            # families' `frac` stays priced against the 8 TB/s spec; `frac_of_probe` prices each against the probe that
            # matches its traffic (write-dominated -> fill, read-dominated -> read, mixed -> copy): how far a family is
            # from what a streaming kernel reaches at all (round-5 verdict item 3: ">= 0.70 or a PMC-backed limit")
            try:
                nel = 157286400 // 4
                pa, pb = torch.empty(nel, device=device), torch.empty(nel, device=device)
                pa.fill_(1.0), pb.copy_(pa), pa.sum()
                torch.cuda.synchronize()

                def gbps(fn, nbytes, reps=50):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    return reps * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9

                probe = {"bytes": nel * 4, "fill_GBps": round(gbps(lambda: pa.fill_(2.0), nel * 4)),
                         "copy_GBps": round(gbps(lambda: pb.copy_(pa), 2 * nel * 4)),
                         "read_GBps": round(gbps(lambda: pa.sum(), nel * 4))}
                del pa, pb
                out["roofline"]["hbm_probe"] = probe
                # the same three passes at 4x the size: beyond the 256 MiB Infinity Cache (a 157 MB fill stays inside it);
                # families whose launches move more than 256 MB are priced against THESE (round 6, tools/hbm_sizes.py)
                nel2 = 4 * nel
                pa, pb = torch.empty(nel2, device=device), torch.empty(nel2, device=device)
                pa.fill_(1.0), pb.copy_(pa), pa.sum()
                torch.cuda.synchronize()
                probe_l = {"bytes": nel2 * 4, "fill_GBps": round(gbps(lambda: pa.fill_(2.0), nel2 * 4, 20)),
                           "copy_GBps": round(gbps(lambda: pb.copy_(pa), 2 * nel2 * 4, 20)),
                           "read_GBps": round(gbps(lambda: pa.sum(), nel2 * 4, 20))}
                del pa, pb
                out["roofline"]["hbm_probe_large"] = probe_l
                kind = {"thin_conv_fwd": "fill", "adam_multi": "copy", "bn_apply": "copy", "bn_bwd_apply": "copy",
                        "upsample2_fwd": "fill", "maxpool2_fwd": "copy", "maxpool2_bwd": "copy", "upsample2_bwd": "copy"}
                for tag, e in hbm.items():
                    if "GBps" in e:
                        k = kind.get(tag, "read")
                        big = e.get("MB_per_launch", 0) > 256
                        e["frac_of_probe"] = round(e["GBps"] / (probe_l if big else probe)[k + "_GBps"], 3)
                        e["probe"] = k + (" (629 MB)" if big else " (157 MB)")
            except Exception as e:
                out["roofline"]["hbm_probe"] = {"failed": repr(e)}
            # the pose critic's k7 TemporalBlock convs (phase3/archis/default.py:207-210; Cout = 128, K = 128 * 7, the
            # weight gradient with its bias column): 5 GFLOP launches that cannot fill 256 CUs x 5 resident workgroups
            #  - their own rate, next to the engine average they pull down
            tcn = [r for r in launches if FAMILIES[r[0]] == "gemm" and
                   ((r[2] == 128 and r[4] == 896) or (r[2] == 128 and r[3] == 897))]
            if tcn:
                t_ms, t_fl = sum(r[5] for r in tcn), sum(r[6] for r in tcn)
                out["roofline"]["tcn_critic"] = {
                    "launches_per_step": round(len(tcn) / args.steps, 1), "ms_per_step": round(t_ms / args.steps, 3),
                    "achieved": round(t_fl / t_ms / 1e9, 1) if t_ms > 0 else 0.0, "unit": "TFLOP/s",
                    "frac": round(t_fl / t_ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4) if t_ms > 0 else 0.0}
            if args.phase == 3:
                # the engine on a plain 4096^3 GEMM, measured here and now (mode 2: both operands K-major, the LDS-direct
                # kernel the conv forward / backward-data launches use; no gather, no tails, a balanced grid): what the
                # engine's own loop and epilogue reach when nothing about the step's shapes is in the way. Standalone
                # kernels reach 140-147 (tools/probes/gemm_ceiling.hip, profiles/r04_gemm_ceiling_*.txt); the round-3
                # "power ceiling of 105" was a misreading. `frac` stays priced against the nominal peak
                n = 4096
                ga, gb = torch.randn(n, n, device=device), torch.randn(n, n, device=device)
                K.gemm(2, ga, gb)
                torch.cuda.synchronize()

                def timed(fn, reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    return reps * 2.0 * n ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12

                # (round 5 timed 10 launches behind the profiling pass: 116 against the probe's 137 for the same kernel - a
                # window too short for the clock to settle; 200 back-to-back launches, ONE event pair, no profiler)
                plain = timed(lambda: K.gemm(2, ga, gb), 200)
                out["roofline"]["plain_gemm_4096_tflops"] = round(plain, 1)
                out["roofline"]["frac_of_plain_gemm"] = round(ach / plain, 4)
                out["roofline"]["plain_gemm_launches_timed"] = 200
                # the same GEMM BETWEEN loop bodies of the step (8 bodies, one launch after each, each launch bracketed by
                # its own event pair): what the engine's plain rate is at the clock / cache state the step leaves behind
                try:
                    evs = []
                    engine.total_iterations = phase
                    for _ in range(8):
                        engine.train_step(*batch)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        K.gemm(2, ga, gb)
                        e1.record()
                        evs.append((e0, e1))
                    engine.flush()
                    torch.cuda.synchronize()
                    ms = sorted(a.elapsed_time(b) for a, b in evs)
                    out["roofline"]["plain_gemm_4096_tflops_between_steps"] = round(2.0 * n ** 3 / (ms[len(ms) // 2] * 1e-3) / 1e12, 1)
                except Exception as e:
                    out["roofline"]["plain_gemm_4096_tflops_between_steps"] = "failed: %r" % (e,)
                # the stand-alone probe's best plain kernel (tools/probes/gemm_ceiling.hip "dl 128x128x16 dma16 frag1"),
                # compiled into the library as a test-only entry point: same process, same 200 launches, and the shader
                # clock it ran at (s_memtime / s_memrealtime over workgroup 0's lifetime)
                try:
                    import ctypes
                    from music2dance_amd import _lib as m2d_lib
                    h = m2d_lib.lib()
                    gc_ = torch.empty(n, n, device=device)
                    st = torch.cuda.current_stream(device).cuda_stream
                    h.m2d_debug_probe_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
                    run_probe = lambda: h.m2d_debug_probe_gemm(ga.data_ptr(), gb.data_ptr(), gc_.data_ptr(), n, n, n, st)
                    run_probe()
                    torch.cuda.synchronize()
                    ref = K.gemm(2, ga, gb)
                    err = float((gc_ - ref).abs().max() / ref.abs().max())
                    out["roofline"]["probe_kernel_tflops"] = round(timed(run_probe, 200), 1)
                    ck = (ctypes.c_ulonglong * 4)()
                    h.m2d_debug_probe_clock(ck)
                    out["roofline"]["probe_kernel_clock_GHz"] = round((ck[3] - ck[1]) / max(ck[2] - ck[0], 1) / 10.0, 3)
                    out["roofline"]["probe_kernel_max_rel_diff_vs_engine"] = float("%.2e" % err)
                    del gc_
                except Exception as e:
                    out["roofline"]["probe_kernel_tflops"] = "failed: %r" % (e,)
                del ga, gb
            if clock is not None:
                out["roofline"]["clock_GHz_in_step"] = clock.get("clock_GHz_in_step")
                out["roofline"]["clock_by_shape"] = clock.get("shapes", clock)
                c = clock.get("clock_GHz_in_step")
                if c:
                    # the fp32-MFMA rate the chip can deliver AT THAT CLOCK (64 FLOP / clock / SIMD x 1024 SIMDs): `frac`
                    # stays priced against the nominal 157.3 (2.4 GHz); this is how much of the gap is clock, not schedule
                    out["roofline"]["peak_at_step_clock_tflops"] = round(PEAK_F32_MFMA_TFLOPS * c / 2.4, 1)
                    out["roofline"]["frac_of_peak_at_step_clock"] = round(ach / (PEAK_F32_MFMA_TFLOPS * c / 2.4), 4)
            # the reference's own formulation executes REF_GFLOP_PER_SEQ_CYCLE per sequence consumed (SURVEY.md
            # 8(d), per workload); the engine skips work the reference discards, so this is an equivalent-work
            # rate over whole cycles, not a kernel rate
            ref_gf = REF_GFLOP_PER_SEQ_CYCLE.get(ref_key)
            if ref_gf is not None and whole_cycles is not None:
                out["roofline"]["reference_formulation"] = {
                    "gflop_per_seq_cycle": ref_gf,
                    "step_tflops": round(ref_gf * 1e9 * whole_cycles["value"] / 1e12, 2)}
        default_cfg = args.phase == 3 and (args.enc_type, args.frames, args.ablated) == ("default", 120, False)
        if not args.no_cpu_baseline and world == 1 and default_cfg:
            pay = None
            try:
                out["cpu_baseline"] = cpu_baseline()
                pay = out["cpu_baseline"].pop("_parity", None)
            except Exception as e:  # the baseline is informational; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "seq/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
            if pay is not None:
                try:
                    out["parity_at_bench_size"] = parity_at_bench_size(pay, device)
                except Exception as e:
                    out["parity_at_bench_size"] = {"failed": repr(e)}
        if args.parity_check and args.phase == 2 and world == 1:
            try:
                out["parity_at_bench_size"] = parity_c2(device, args.batch, args.frames)
            except Exception as e:
                out["parity_at_bench_size"] = {"failed": repr(e)}
        if others is not None:
            out["other_configs"] = others
        print(json.dumps(out), flush=True)
    watchdog.cancel()
    if faults_timed:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit("bench.py: %d persistent-kernel timeout(s) inside the timed window: optimizer steps were voided, the "
                 "rate above is not a training rate" % faults_timed)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
