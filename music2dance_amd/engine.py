"""WGAN-GP training engine: the inner loops of the reference's train scripts
(phase1/train_wgan-gp.py:79-110, phase2/train.py:135-180, phase3/train.py:186-243) as
reusable step functions, device-resident and free of per-iteration host syncs.

Results follow the reference's formulation; work the reference computes and discards is
skipped: the critic-iteration generator forward keeps no autograd graph, the audio branch
of the phase-3 critic is evaluated once per iteration (SequenceDiscriminator.shared_audio),
the raw-audio gradient is not formed outside the penalty, and during generator steps the
critic's parameters are frozen instead of receiving gradients that the next zero_grad
throws away.
"""
import contextlib
import os

import torch
import torch.optim as optim

from . import kernels, ops
from .critic_step import CriticStep
from .dp import GradExchange
from .layers import DrawTape, copy_stream, draw_tape, host_draw, to_device_async
from .losses import gradient_penalty, tv_loss


class _Freeze:
    """Temporarily mark parameters as not requiring grad (generator steps)."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)


def _capture(graph, body, dev, keep, pool=None):
    """Capture body() into `graph` on a capture stream of its own, with kernel scratch (split-K arrival counters,
    BatchNorm accumulators) that belongs to this graph alone: captured launches bake the scratch address in, and graphs
    that replay concurrently (the phase-2 generator forward under the previous body's critic pass) must not count
    arrivals in each other's words - torch's default is ONE capture stream for every graph, hence one scratch.
    `keep`: a list that outlives the graph; stream and scratch are parked there."""
    cap = torch.cuda.Stream(device=dev)
    scratch = kernels.private_scratch(dev)
    keep.append((cap, scratch))
    # capture_error_mode="thread_local": with the default ("global") ANY thread's HIP call that is illegal during a
    # capture fails - and RCCL's process-group watchdog thread polls the events of pending collectives all the time: one
    # run in five of the data-parallel graph test died with "operation not permitted when stream is capturing" raised in
    # that thread (round 5, tests/test_gpu_dp.py looped). Only this thread captures; the others are none of its business.
    with scratch, torch.cuda.graph(graph, pool=pool, stream=cap, capture_error_mode="thread_local"):
        return body()



def timed_wait(stream, event, pairs):
    """stream.wait_event(event), bracketed by two timing events on `stream` (nothing else between them): the pair goes to
    `pairs` (bounded) - how long the stream sat waiting. Not under graph capture."""
    if pairs is None or torch.cuda.is_current_stream_capturing():
        stream.wait_event(event)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    stream.wait_event(event)
    e1.record(stream)
    pairs.append((e0, e1))
    if len(pairs) > 512:
        del pairs[:256]

class WganGpEngine:
    """Common machinery: optimisers, n_critic gating, data-parallel gradient exchange (the critic's
    optimiser step is taken at the start of the next iteration, once its all-reduce has landed),
    the critic iterations' generator forward one iteration ahead on a second stream."""

    def __init__(self, gen, critic, lr_gen, lr_critic, n_critic_steps, data_parallel=True, fused_adam=None,
                 sync_bn=False):
        self.gen, self.critic = gen, critic
        # sync_bn: BatchNorm statistics over the global batch (all-reduced sums) instead of per rank
        self.sync_bn = bool(sync_bn) and ops.set_sync_batchnorm(True)
        self.n_critic_steps = int(n_critic_steps)
        dev = next(critic.parameters()).device
        # torch.optim.Adam objects (param_groups, state_dict, schedulers) whose step is ONE multi-tensor HIP launch that
        # also rewrites the packed images of the conv weights it changes (optim.py; M2D_ADAM=torch: torch's own fused /
        # foreach implementation - the A/B lever; `fused_adam` then keeps its round-1 meaning)
        if os.environ.get("M2D_ADAM", "m2d") == "torch":
            kw = {}
            if fused_adam is None:
                fused_adam = dev.type == "cuda"
            if fused_adam:
                kw["fused"] = True
            self.optim_critic = optim.Adam(critic.parameters(), lr=lr_critic, **kw)
            self.optim_gen = optim.Adam(gen.parameters(), lr=lr_gen, **kw)
        else:
            from .optim import Adam as M2dAdam
            self.optim_critic = M2dAdam(critic.parameters(), lr=lr_critic)
            self.optim_gen = M2dAdam(gen.parameters(), lr=lr_gen)
        # an optimizer step only invalidates the packed conv-weight images of ITS parameters (the generator's
        # stay valid through the critic iterations of a cycle)
        self._keep_packs = os.environ.get("M2D_KEEP_PACKS", "1") != "0"
        self._critic_params = [p for p in critic.parameters() if p.dim() == 3] if self._keep_packs else None
        # (the generator's 2-D GRU weights too: their transposes are cached the same way)
        self._gen_params = [p for p in gen.parameters() if p.dim() in (2, 3)] if self._keep_packs else None
        self.total_iterations = 0
        # the critic's exchange (every iteration) starts bucket by bucket underneath its own backward pass
        self.x_critic = GradExchange(critic.parameters()).overlap_backward() if data_parallel else None
        # the generator's exchange (every n_critic-th iteration) leaves in 4 MB buckets from the hooks of its backward
        # pass (one autograd backward per generator iteration): only the last bucket's ring is exposed
        self.x_gen = GradExchange(gen.parameters(), bucket_mb=4.0).overlap_backward() if data_parallel else None
        self._critic_step_pending = False
        # a persistent recurrent launch that times out must not take the run with it (_check_async): both optimizers
        # read the fault word - directly, or (data parallel) as reduced over the ranks by their gradient exchange
        k = kernels.impl()
        if hasattr(self.optim_critic, "skip_flag") and hasattr(k, "fault_word") and dev.type == "cuda":
            for opt, x in ((self.optim_critic, self.x_critic), (self.optim_gen, self.x_gen)):
                opt.skip_flag = x.fault_flag(k.fault_fetch) if (x is not None and x.active) else k.fault_word()
        # the critic iteration as one hand-scheduled pass (critic_step.py) instead of three autograd passes; engines
        # whose critic qualifies (piecewise-linear heads) build it in their constructor. M2D_MANUAL_CRITIC=0: autograd
        self.manual_critic = None
        self.last = {}
        self.last_full = {}  # most recent value of every scalar (generator scalars persist between G steps)
        # generator-forward pipelining (see _generator_forward_nograd)
        self.pipeline_generator = os.environ.get("M2D_GEN_PIPELINE", "1") != "0"
        self._inputs_ready = None
        self._gen_stream = None
        self._gen_params_ready = None
        self._main_mark = None
        self._fake_pending = None
        # diagnostics: how long the consumer of a pipelined generator forward waited for it (event pairs on the waiting
        # stream; join_stats()). With data parallelism this and GradExchange.wait_stats() are what an efficiency loss
        # at N > 1 GPUs is attributable to.
        self._join_pairs = []
        self._arm_keep = False     # set by train_step for the body that holds a generator iteration
        self._gen_epoch = 0        # bumped by every generator optimizer step (fused steps do not move version counters)
        self._kept_key = None

    # -- critic optimiser step, possibly deferred so its all-reduce overlaps the next G forward
    def _finish_critic_step(self):
        if self._critic_step_pending:
            if self.x_critic is not None:
                self.x_critic.finish()
            self.optim_critic.step()
            kernels.impl().invalidate_packed(self._critic_params)
            self._critic_step_pending = False

    def _begin_critic_step(self):
        if self.x_critic is not None and self.x_critic.active:
            self.x_critic.start()
            self._critic_step_pending = True
        else:
            self.optim_critic.step()
            kernels.impl().invalidate_packed(self._critic_params)

    def _poll_exchange(self):
        x = self.x_critic
        if x is None or not x.active or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            return None
        return x.poll

    def _gen_step(self):
        if self.x_gen is not None:
            self.x_gen.exchange()
        self.optim_gen.step()
        self._gen_epoch += 1
        kernels.impl().invalidate_packed(self._gen_params)
        if self._gen_stream is not None:
            self._gen_params_ready = torch.cuda.current_stream().record_event()

    def _generator_forward_nograd(self, fn, inputs, device=None, grad=False):
        """The generator forward of a critic iteration (no autograd graph: the reference builds one and
        drops it). It reads the generator's weights and the batch, nothing the critic's optimizer
        touches, so it need not wait for the previous critic iteration: when the caller passes
        `inputs_ready` (an event after which the batch tensors are complete - a copy stream's, or one
        recorded before the loop for resident data) it runs on a second stream and the launch-bound
        part of the generator (the recurrent layers, the decoder's small layers) executes underneath
        the previous iteration's critic GEMMs. Same kernels, same operands, same results; without
        `inputs_ready` the forward runs in line."""
        ready = self._inputs_ready
        dev = device if device is not None else inputs[0].device
        # grad: this forward's audio path is kept (with its autograd graph) for the generator iteration of the same
        # loop body (Phase3Engine.reuse_audio_path); everything else about it is as without
        mode = torch.enable_grad if grad else torch.no_grad
        if ready is None or not self.pipeline_generator or dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
            with mode():
                return fn()
        main = torch.cuda.current_stream(dev)
        if self._gen_stream is None:
            self._gen_stream = torch.cuda.Stream(dev)  # default priority: a high-priority queue preempts (measured +25 %)
            self._gen_params_ready = main.record_event()  # weights as initialised / loaded on the main stream
            # (float tensors only: the BatchNorm step counters are bumped by every forward itself)
            self._gen_tensors = [t for t in list(self.gen.parameters()) + list(self.gen.buffers()) if t.is_floating_point()]
            self._gen_versions = sum(t._version for t in self._gen_tensors)
        gs = self._gen_stream
        # parameters / buffers changed by ordinary tensor operations since the last forward (load_state_dict, manual
        # edits - they move the version counters; the engine's own fused optimizer step is covered in _gen_step):
        # those ran on the main stream, so this forward has to queue behind it once
        ver = sum(t._version for t in self._gen_tensors)
        if ver != self._gen_versions:
            self._gen_versions = ver
            self._gen_params_ready = main.record_event()
        gs.wait_event(ready)
        gs.wait_event(self._gen_params_ready)
        if self._main_mark is not None:
            # at most one iteration ahead: not before the main stream has taken delivery of the previous result
            gs.wait_event(self._main_mark)
        with torch.cuda.stream(gs), mode():
            out = fn()
            done = gs.record_event()
        for t in inputs:
            t.record_stream(gs)
        self._fake_pending = (out, done)  # the consumer calls _join_generator_forward before reading `out`
        return out

    def _join_generator_forward(self, out):
        pend, self._fake_pending = self._fake_pending, None
        if pend is None:
            return
        assert pend[0] is out
        main = torch.cuda.current_stream(out.device)
        timed_wait(main, pend[1], self._join_pairs)
        self._main_mark = main.record_event()
        out.record_stream(main)

    def join_stats(self, reset=False):
        """-> {"joins", "wait_ms_mean", "wait_ms_max"}: time the consuming stream waited for a generator forward that ran
        on the generator stream. Call with the device idle."""
        vals = [a.elapsed_time(b) for a, b in self._join_pairs if b.query()]
        if reset:
            del self._join_pairs[:]
        if not vals:
            return {"joins": 0, "wait_ms_mean": None, "wait_ms_max": None}
        return {"joins": len(vals), "wait_ms_mean": round(sum(vals) / len(vals), 4), "wait_ms_max": round(max(vals), 4)}

    def _hand_over_generator_forward(self, out):
        """As _join_generator_forward, but the main stream does not wait: -> the completion event (or None) for the
        consumer to wait on where it first reads `out` (CriticStep.run: its pose branch)."""
        pend, self._fake_pending = self._fake_pending, None
        if pend is None:
            return None
        assert pend[0] is out
        main = torch.cuda.current_stream(out.device)
        self._main_mark = main.record_event()
        out.record_stream(main)
        return pend[1]

    def flush(self):
        self._finish_critic_step()
        # the persistent recurrent launches report a timeout through a host word that is only meaningful once the
        # device has caught up: synchronise here (end of an epoch / before a checkpoint / end of the bench), not per step
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()
        self._check_async()

    def _check_async(self):
        """A persistent recurrent launch that gave up waiting for its peers (CUs held by another resident kernel - an
        RCCL collective, another process) raises a word in pinned host memory; reading it costs nothing. Round 3 died
        here. Now: the optimizers take that word as their step's `skip` flag (optim.Adam / m2d_adam_multi), so every
        update queued behind the failed launch voids itself ON THE DEVICE; the host, finding the word raised,
        synchronises, clears it and carries on with per-step recurrent launches (kernels.recover_async_fault). The
        affected iterations are lost, not the run; `async_faults` counts them. Data-parallel ranks must skip together:
        there the word travels with every gradient exchange (one more float in the last bucket, dp.GradExchange.fault_flag),
        and the REDUCED value is what the optimizers read."""
        k = kernels.impl()
        rec = getattr(k, "recover_async_fault", None)
        if rec is None:
            return
        if rec():
            import warnings
            warnings.warn("a persistent recurrent launch timed out: its iteration(s) were skipped on the device, the "
                          "recurrences run as per-step launches from here on (%d so far)" % k.async_faults)
        # Captured graphs have the persistent recurrent kernels baked in: after a recovery (by this engine or by another
        # engine of the process - the switch to per-step launches is process-wide) every replay would run them again and
        # could time out again, voiding iterations for ever. Graphs captured before the latest recovery are dropped
        # (with their private scratch, which a killed launch may have left non-zero); the next train_step re-captures,
        # now with per-step recurrent launches.
        gen = int(getattr(k, "async_faults", 0))
        if getattr(self, "_graphs", None) and getattr(self, "_graphs_gen", gen) != gen:
            self._graphs = {}
        self._graphs_gen = gen

    def train_step(self, *batch, inputs_ready=None):
        """One loop body of the reference: a critic iteration, plus a generator iteration every
        n_critic_steps-th call. Returns a dict of 0-dim device tensors (no host sync).
        inputs_ready: optional event after which the batch tensors are complete; lets the generator
        forward start before the previous iteration has drained (_generator_forward_nograd)."""
        self.total_iterations += 1
        self._inputs_ready = inputs_ready
        self._check_async()
        with_gen = self.total_iterations % self.n_critic_steps == 0
        # this body's critic iteration may keep the generator's audio path for the generator iteration that follows IN
        # THIS CALL (Phase3Engine.reuse_audio_path): armed here and nowhere else - callers that drive critic_iteration /
        # generator_iteration themselves never keep anything (inferring it from total_iterations made every
        # stand-alone critic iteration keep a graph that a later generator iteration on ANOTHER batch would consume)
        self._arm_keep = with_gen
        try:
            # weights only change in the optimizer steps, which drop the packed conv-weight images
            with kernels.impl().weight_cache(keep=self._keep_packs):
                out = self.critic_iteration(*batch)
                if with_gen:
                    out.update(self.generator_iteration(*batch))
        finally:
            self._arm_keep = False
        self.last = out
        self.last_full.update(out)
        return out


# =========================================================================================== phase 3
class Phase3Engine(WganGpEngine):
    """Audio-conditioned sequence WGAN-GP (phase3/train.py:186-243)."""

    def __init__(self, gen, critic, cfg, ablated=False, **kw):
        super().__init__(gen, critic, cfg["lr_gen"], cfg["lr_critic"], cfg["n_critic_steps"], **kw)
        self.gamma, self.beta, self.eta = float(cfg["gamma"]), float(cfg["beta"]), float(cfg["eta"])
        self.output_size = int(cfg.get("output_size", 69))
        self.ablated = bool(ablated)
        self.early_pair_pass = os.environ.get("M2D_EARLY_PAIR", "1") != "0"
        # The loop body that holds a generator iteration runs the generator TWICE on one batch (phase3/train.py:195 for the
        # critic iteration, :222 for the generator iteration, fresh noise): encoder and audio GRU do not see the noise, so
        # the second pass's audio path equals the first's. With this on, the first pass of such a body keeps it (autograd
        # graph included) and the generator iteration runs only noise GRU + decoder on top, replaying the encoder's
        # BatchNorm running-statistics update with the same sums (SequenceGenerator.forward_keeping_audio_path).
        # M2D_REUSE_AUDIO_PATH=0: both passes in full, as the reference executes them.
        self.reuse_audio_path = (os.environ.get("M2D_REUSE_AUDIO_PATH", "1") != "0"
                                 and hasattr(gen, "forward_keeping_audio_path"))
        if os.environ.get("M2D_MANUAL_CRITIC", "1") != "0" and CriticStep.supports(critic):
            self.manual_critic = CriticStep(critic, self.gamma, lp=False)
            self.manual_critic.join_pairs = self._join_pairs

    def _shapes(self, real):
        B = real.size(0)
        T = real.numel() // (B * self.output_size)
        return B, T

    def _audio_path_key(self, audio_slices):
        """Identity of (batch, generator weights) a kept audio path is valid for: the batch tensor's storage, layout and
        version counter; the generator's optimizer-step count and the version counters of its parameters and float
        buffers (load_state_dict and manual edits move those; the fused optimizer step moves the count)."""
        vers = sum(t._version for t in list(self.gen.parameters()) + [b for b in self.gen.buffers() if b.is_floating_point()])
        return (audio_slices.data_ptr(), tuple(audio_slices.shape), tuple(audio_slices.stride()), audio_slices._version,
                self._gen_epoch, vers)

    def critic_iteration(self, real, audio, audio_slices):
        """real (B, T, 69) [any view of B*T*69], audio (B, samples), audio_slices (B, T, window)."""
        out = self._critic_body(real, audio, audio_slices, None, None, True)
        self._begin_critic_step()
        return out

    def _critic_body(self, real, audio, audio_slices, noise, alpha, finish_inside):
        """Forward / backward of the critic iteration up to the gradients (no optimizer step).
        noise / alpha: None = drawn from the host generator where the reference draws them."""
        B, T = self._shapes(real)
        # the reference builds and drops this graph (phase3/train.py:195)
        keep = (self.reuse_audio_path and finish_inside and self.gen.training and self._arm_keep
                and not (real.is_cuda and torch.cuda.is_current_stream_capturing()))
        self._kept_key = None
        if keep:
            fake_rows = self._generator_forward_nograd(
                lambda: self.gen.forward_keeping_audio_path(audio_slices, [T] * B, noise).detach(), (audio_slices,), grad=True)
            # what the kept path belongs to: THIS batch tensor as it is now, THESE generator weights
            self._kept_key = self._audio_path_key(audio_slices)
            # the forward may have run on the generator stream (one body ahead): its completion event orders the main
            # stream's reads of the kept tensors in the generator iteration (forward_from_kept_audio_path)
            self._kept_done = self._fake_pending[1] if self._fake_pending is not None else None
        else:
            if hasattr(self.gen, "drop_kept_audio_path"):
                self.gen.drop_kept_audio_path()
            fake_rows = self._generator_forward_nograd(lambda: self.gen(audio_slices, [T] * B, noise), (audio_slices,))
        if finish_inside:
            self._finish_critic_step()
        # only after the deferred step has consumed the previous iteration's gradients
        self.optim_critic.zero_grad(set_to_none=True)
        audio_c = audio.unsqueeze(1)
        if self.manual_critic is not None:
            # the audio branch's forward does not read the poses: after a generator step, when the generator forward
            # cannot run ahead, it starts while that forward is still going; only the pose branch waits for it
            fake_ready = self._hand_over_generator_forward(fake_rows)
            if alpha is None:  # drawn on the host generator where the reference draws it (losses.py:15)
                alpha = to_device_async(torch.rand(B, 1), real.device)
            return self.manual_critic.run(real, fake_rows, None if self.ablated else audio_c, alpha,
                                          on_grads=self._poll_exchange(), fake_ready=fake_ready)
        real_c = real.view(B, T, self.output_size).permute(0, 2, 1).contiguous()
        with self.critic.shared_audio() if not self.ablated else contextlib.nullcontext():
            self._join_generator_forward(fake_rows)
            return self._critic_passes(B, T, real, real_c, fake_rows, audio_c, alpha)

    def _critic_passes(self, B, T, real, real_c, fake_rows, audio_c, alpha):
        fake = fake_rows.view(B, T, self.output_size).permute(0, 2, 1).contiguous()
        if self.ablated:
            gp = gradient_penalty(self.critic, B, real_c, fake, is_seq=True, lp=False, device=real.device, alpha=alpha)
            s_real, s_fake = self.critic.score_pair(real_c, fake)
            err_real, err_fake = s_real.mean(), s_fake.mean()
            err_critic = err_fake - err_real + self.gamma * gp
            err_critic.backward()
        else:
            if self.early_pair_pass:
                self.critic.begin_pair(real_c, fake)  # pose branch of the real / fake pass: see begin_pair
            gp = gradient_penalty(self.critic, B, real_c, fake, audio_c, is_seq=True, lp=False,
                                  device=real.device, alpha=alpha)
            s_real, s_fake = self.critic.score_pair(real_c, fake, audio_c)
            err_real, err_fake = s_real.mean(), s_fake.mean()
            err_critic = err_fake - err_real + self.gamma * gp
            with ops.no_input_grad_for(audio_c):
                err_critic.backward()
            audio_c.requires_grad_(False)
            if getattr(self.critic, "_stick_stream", None) is not None:
                # the pose branch's gradients were allocated on the critic's side stream; the optimizer reads them on
                # this one (autograd has made it wait for them, but the allocator does not know about the read)
                cur = torch.cuda.current_stream(real.device)
                for p in self.critic.stick_d.parameters():
                    if p.grad is not None:
                        p.grad.record_stream(cur)
        return {"loss_critic": err_critic.detach(), "gp": gp.detach(), "w_dist": (err_fake - err_real).detach()}

    def generator_iteration(self, real, audio, audio_slices):
        self._finish_critic_step()
        out = self._generator_body(real, audio, audio_slices, None)
        self._gen_step()
        return out

    def _generator_body(self, real, audio, audio_slices, noise):
        B, T = self._shapes(real)
        self.optim_gen.zero_grad(set_to_none=True)
        use_kept = (self.reuse_audio_path and self.gen.kept_audio_path() and self.gen.training
                    and self._kept_key is not None and self._kept_key == self._audio_path_key(audio_slices))
        self._kept_key = None
        if use_kept:
            # (the critic iteration of this body kept it: same batch tensor, same weights - checked, not assumed)
            fake_rows = self.gen.forward_from_kept_audio_path(noise, after=getattr(self, "_kept_done", None))
        else:
            if hasattr(self.gen, "drop_kept_audio_path"):
                self.gen.drop_kept_audio_path()
            fake_rows = self.gen(audio_slices, [T] * B, noise)
        self._kept_done = None
        real_rows = real.reshape(B * T, self.output_size)
        err_l1 = ops.l1_mean(real_rows, fake_rows)
        fake = fake_rows.view(B, T, self.output_size).permute(0, 2, 1)
        real_c = real.view(B, T, self.output_size).permute(0, 2, 1).contiguous()
        audio_c = audio.unsqueeze(1)
        with _Freeze(self.critic):
            if self.ablated:
                with torch.no_grad():
                    err_real = self.critic(real_c).mean()
                err_fake = self.critic(fake).mean()
            else:
                with self.critic.shared_audio():
                    with torch.no_grad():
                        err_real = self.critic(real_c, audio_c).mean()
                    err_fake = self.critic(fake, audio_c).mean()
            err_tv = tv_loss(fake)
            err_gen = err_real - err_fake + self.beta * err_l1 + self.eta * err_tv
            err_gen.backward()
        return {"loss_gen": err_gen.detach(), "l1_loss_train": err_l1.detach()}

    # ------------------------------------------------------------------ captured-graph mode
    def enable_graphs(self, on=True):
        """Replay each loop body's forward / backward as one captured HIP graph (per input shape):
        ~680 kernel launches become one graph launch, which takes the host off the critical path.
        (Measured on the MI355X host: batch 8, 8.5 ms per body eager -> 5.9 ms replayed; at batch 64 the eager
        path is GPU-bound and, with the generator forward pipelined one iteration ahead, faster than
        the single-queue replay - 15.0 vs 16.5 ms - so this is off by default: it is for small
        batches and slow hosts.)
        The optimizer steps and the data-parallel gradient exchange stay eager. The random draws
        the reference makes inside the loop (generator noise, interpolation weights) are made on
        the host generator in the same order before each replay and fed through static buffers,
        so results equal the eager path's."""
        self._use_graphs = bool(on)
        self._graphs = {}
        if on and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            # warm-up and capture run on side streams by design
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        return self

    def train_step(self, real, audio, audio_slices, inputs_ready=None):
        if not getattr(self, "_use_graphs", False) or real.device.type != "cuda":
            return super().train_step(real, audio, audio_slices, inputs_ready=inputs_ready)
        self.total_iterations += 1
        self._check_async()
        self._finish_critic_step()
        g = self._graph_for(real, audio, audio_slices)
        B, T = self._shapes(real)
        nz = self.gen.noise_size
        g["real"].copy_(real.reshape(g["real"].shape))
        g["audio"].copy_(audio)
        g["slices"].copy_(audio_slices)
        # host draws in the reference's order: generator noise, then the penalty's alpha
        g["noise_c"].copy_(to_device_async(torch.randn(B, T, nz), real.device))
        g["alpha"].copy_(to_device_async(torch.rand(B, 1), real.device))
        # every captured graph writes its gradients into the tensors it was captured with
        self._bind_grads(self.critic, g["critic_grads"])
        g["critic"].replay()
        self._begin_critic_step()
        out = dict(g["critic_out"])
        if self.total_iterations % self.n_critic_steps == 0:
            self._finish_critic_step()
            g["noise_g"].copy_(to_device_async(torch.randn(B, T, nz), real.device))
            self._bind_grads(self.gen, g["gen_grads"])
            g["gen"].replay()
            self._gen_step()
            out.update(g["gen_out"])
        self.last = out
        self.last_full.update(out)
        return out

    @staticmethod
    def _bind_grads(module, grads):
        for p, gr in zip(module.parameters(), grads):
            p.grad = gr

    def _graph_for(self, real, audio, audio_slices):
        key = (tuple(real.shape), tuple(audio.shape), tuple(audio_slices.shape))
        g = self._graphs.get(key)
        if g is not None:
            return g
        dev = real.device
        B, T = self._shapes(real)
        nz = self.gen.noise_size
        g = {"real": torch.empty_like(real).copy_(real), "audio": torch.empty_like(audio).copy_(audio),
             "slices": torch.empty_like(audio_slices).copy_(audio_slices),
             "noise_c": torch.zeros(B, T, nz, device=dev), "noise_g": torch.zeros(B, T, nz, device=dev),
             "alpha": torch.full((B, 1), 0.5, device=dev)}
        K = kernels.impl()

        def critic_body():
            with K.weight_cache():
                return self._critic_body(g["real"], g["audio"], g["slices"], g["noise_c"], g["alpha"], False)

        def gen_body():
            with K.weight_cache():
                return self._generator_body(g["real"], g["audio"], g["slices"], g["noise_g"])

        # One warm-up pass of each body on a side stream (lazy module loads and allocator growth
        # must not happen inside a capture); it must leave no trace: the BatchNorm buffers it
        # touches are restored, gradients are dropped, no optimizer step is taken, no host draw.
        mods = [self.gen, self.critic]
        saved = [[b.clone() for b in m.buffers()] for m in mods]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # (the warm-up backward must not feed the data-parallel exchange either: no start() follows it)
        quiet = [x.suspended() for x in (self.x_critic, self.x_gen) if x is not None]
        with torch.cuda.stream(side), contextlib.ExitStack() as es:
            for q in quiet:
                es.enter_context(q)
            critic_body()
            gen_body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.optim_critic.zero_grad(set_to_none=True)
        self.optim_gen.zero_grad(set_to_none=True)
        g["keep"] = []
        g["critic"] = torch.cuda.CUDAGraph()
        g["critic_out"] = _capture(g["critic"], critic_body, dev, g["keep"])
        # the gradient tensors live in this graph's private pool: a graph captured later for another
        # shape allocates its own, so each replay re-binds p.grad to the set it writes
        g["critic_grads"] = [p.grad for p in self.critic.parameters()]
        g["gen"] = torch.cuda.CUDAGraph()
        g["gen_out"] = _capture(g["gen"], gen_body, dev, g["keep"], pool=g["critic"].pool())
        g["gen_grads"] = [p.grad for p in self.gen.parameters()]
        with torch.no_grad():
            for m, bufs in zip(mods, saved):
                for b, v in zip(m.buffers(), bufs):
                    b.copy_(v)
        torch.cuda.synchronize(dev)
        self._graphs[key] = g
        return g

    @torch.no_grad()
    def validation_l1(self, batches):
        """Eval-mode L1 validation loss (phase3/train.py:245-261): mean over batches of
        L1(real, gen(audio_slices)) with the generator's running BatchNorm statistics."""
        was_training = self.gen.training
        self.gen.eval()
        vals = []
        for real, audio_slices in batches:
            B = real.size(0)
            T = real.numel() // (B * self.output_size)
            fake_rows = self.gen(audio_slices, [T] * B)
            vals.append(ops.l1_mean(real.reshape(B * T, self.output_size).contiguous(), fake_rows))
        self.gen.train(was_training)
        return torch.stack(vals).mean()


# =========================================================================================== phase 2
class Phase2Engine(WganGpEngine):
    """Unconditional sequence WGAN-LP (phase2/train.py:135-180), with the reference's
    MultiStepLR schedulers stepped on generator iterations only (:179-180)."""

    def __init__(self, gen, critic, cfg, **kw):
        super().__init__(gen, critic, cfg["lr_gen"], cfg["lr_critic"], cfg["n_critic_steps"], **kw)
        self.gamma, self.eta = float(cfg["gamma"]), float(cfg["eta"])
        self.input_size = int(cfg["input_vector_size"])
        self.output_size = int(cfg.get("output_size", 69))
        ms = [10000, 35000, 50000]
        self.scheduler_critic = optim.lr_scheduler.MultiStepLR(self.optim_critic, milestones=ms, gamma=0.8)
        self.scheduler_gen = optim.lr_scheduler.MultiStepLR(self.optim_gen, milestones=ms, gamma=0.8)
        self.host_noise = True  # draw noise on the host generator (matches the CPU reference)
        if os.environ.get("M2D_MANUAL_CRITIC", "1") != "0" and CriticStep.supports(critic):
            self.manual_critic = CriticStep(critic, self.gamma, lp=True)
            self.manual_critic.join_pairs = self._join_pairs

    def _noise(self, B, T, device):
        if self.host_noise:
            return to_device_async(torch.randn(B, T, self.input_size), device)
        return torch.randn(B, T, self.input_size, device=device)

    def _always_ready(self, device):
        """The generator's only input is noise the engine draws itself: its forward may always run one iteration
        ahead on the second stream (an event that has long completed plays the loader's `inputs_ready`)."""
        if self._inputs_ready is None and device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            if getattr(self, "_ready0", None) is None:
                self._ready0 = torch.cuda.current_stream(device).record_event()
            self._inputs_ready = self._ready0

    def critic_iteration(self, real):
        out = self._critic_body(real, None, None)
        self._begin_critic_step()
        return out

    def _critic_body(self, real, noise, alpha):
        """noise / alpha: None = drawn where the reference draws them (phase2/train.py:139-140, losses.py:15)."""
        B = real.size(0)
        T = real.numel() // (B * self.output_size)
        self._always_ready(real.device)
        # (the noise is drawn inside the forward: on the generator's stream when it runs ahead)
        fake_rows = self._generator_forward_nograd(
            lambda: self.gen(self._noise(B, T, real.device) if noise is None else noise, [T] * B), (), real.device)
        self._finish_critic_step()
        self.optim_critic.zero_grad(set_to_none=True)  # after the deferred step used the old gradients
        self._join_generator_forward(fake_rows)
        return self._critic_from_fake(real, fake_rows, alpha)

    def _critic_from_fake(self, real, fake_rows, alpha):
        B = real.size(0)
        T = real.numel() // (B * self.output_size)
        if self.manual_critic is not None:
            if alpha is None:
                alpha = to_device_async(torch.rand(B, 1), real.device)  # host draw, as losses.py:15
            return self.manual_critic.run(real, fake_rows, None, alpha, on_grads=self._poll_exchange())
        fake = fake_rows.view(B, T, self.output_size).permute(0, 2, 1).contiguous()
        real_c = real.view(B, T, self.output_size).permute(0, 2, 1).contiguous()
        gp = gradient_penalty(self.critic, B, real_c, fake, is_seq=True, lp=True, device=real.device, alpha=alpha)
        s_real, s_fake = self.critic.score_pair(real_c, fake)
        err_real, err_fake = s_real.mean(), s_fake.mean()
        err_critic = err_fake - err_real + self.gamma * gp
        err_critic.backward()
        return {"loss_critic": err_critic.detach(), "gp": gp.detach(), "w_dist": (err_fake - err_real).detach()}

    def generator_iteration(self, real):
        self._finish_critic_step()
        out = self._generator_body(real, None)
        self._gen_step()
        self.scheduler_critic.step()
        self.scheduler_gen.step()
        return out

    def _generator_body(self, real, noise):
        B = real.size(0)
        T = real.numel() // (B * self.output_size)
        self.optim_gen.zero_grad(set_to_none=True)
        if noise is None:
            noise = self._noise(B, T, real.device)
        fake = self.gen(noise, [T] * B).view(B, T, self.output_size).permute(0, 2, 1)
        real_c = real.view(B, T, self.output_size).permute(0, 2, 1).contiguous()
        with _Freeze(self.critic):
            with torch.no_grad():
                err_real = self.critic(real_c).mean()
            err_fake = self.critic(fake).mean()
            err_gen = err_real - err_fake + self.eta * tv_loss(fake)
            err_gen.backward()
        return {"loss_gen": err_gen.detach()}

    # ------------------------------------------------------------------ captured-graph mode (as Phase3Engine's)
    def enable_graphs(self, on=True):
        """Replay each loop body's forward / backward as one captured HIP graph per batch shape. The optimizer and
        scheduler steps and the gradient exchange stay eager; the noise and the interpolation weights are drawn
        where the eager path draws them, in the same order, and fed through static buffers."""
        self._use_graphs = bool(on)
        self._graphs = {}
        if on and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        return self

    def train_step(self, real, inputs_ready=None):
        if not getattr(self, "_use_graphs", False) or real.device.type != "cuda":
            return super().train_step(real, inputs_ready=inputs_ready)
        self.total_iterations += 1
        self._check_async()
        self._finish_critic_step()
        g = self._graph_for(real)
        B, T = g["noise_c"].shape[0], g["noise_c"].shape[1]
        dev = real.device
        # the critic iteration's generator forward is a graph of its own, replayed on the generator stream one
        # iteration ahead (as the eager engine pipelines it): a single replay queue would put its 0.5 ms of recurrent
        # launches in line with the critic (2.64 ms per body against 2.2 eager)
        self._always_ready(dev)

        def forward():
            g["noise_c"].copy_(self._noise(B, T, dev))  # host draw where the eager path makes it
            g["fwd"].replay()
            return g["fake_out"]

        out_rows = self._generator_forward_nograd(forward, (), dev)
        pend, self._fake_pending = self._fake_pending, None
        main = torch.cuda.current_stream(dev)
        g["real"].copy_(real.reshape(g["real"].shape))
        g["alpha"].copy_(to_device_async(torch.rand(B, 1), dev))
        if pend is not None:
            main.wait_event(pend[1])
        g["fake_in"].copy_(out_rows)
        if pend is not None:
            # the next forward may overwrite its output only after this copy
            self._main_mark = main.record_event()
        Phase3Engine._bind_grads(self.critic, g["critic_grads"])
        g["critic"].replay()
        self._begin_critic_step()
        out = dict(g["critic_out"])
        if self.total_iterations % self.n_critic_steps == 0:
            self._finish_critic_step()
            g["noise_g"].copy_(self._noise(B, T, real.device))
            Phase3Engine._bind_grads(self.gen, g["gen_grads"])
            g["gen"].replay()
            self._gen_step()
            self.scheduler_critic.step()
            self.scheduler_gen.step()
            out.update(g["gen_out"])
        self.last = out
        self.last_full.update(out)
        return out

    def _graph_for(self, real):
        key = tuple(real.shape)
        g = self._graphs.get(key)
        if g is not None:
            return g
        dev = real.device
        B = real.size(0)
        T = real.numel() // (B * self.output_size)
        g = {"real": torch.empty_like(real).copy_(real), "noise_c": torch.zeros(B, T, self.input_size, device=dev),
             "noise_g": torch.zeros(B, T, self.input_size, device=dev), "alpha": torch.full((B, 1), 0.5, device=dev),
             "fake_in": torch.zeros(B * T, self.output_size, device=dev)}
        K = kernels.impl()

        def fwd_body():
            with K.weight_cache(), torch.no_grad():
                return self.gen(g["noise_c"], [T] * B)

        def critic_body():
            self.optim_critic.zero_grad(set_to_none=True)
            with K.weight_cache():
                return self._critic_from_fake(g["real"], g["fake_in"], g["alpha"])

        def gen_body():
            with K.weight_cache():
                return self._generator_body(g["real"], g["noise_g"])

        mods = [self.gen, self.critic]
        saved = [[b.clone() for b in m.buffers()] for m in mods]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        quiet = [x.suspended() for x in (self.x_critic, self.x_gen) if x is not None]
        with torch.cuda.stream(side), contextlib.ExitStack() as es:
            for q in quiet:
                es.enter_context(q)
            fwd_body()
            critic_body()
            gen_body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.optim_critic.zero_grad(set_to_none=True)
        self.optim_gen.zero_grad(set_to_none=True)
        # the forward graph replays on the generator stream WHILE the critic graph of the previous body runs: it gets
        # a memory pool of its own (graphs that share a pool reuse each other's freed intermediates, which is only
        # safe when they never overlap - sharing one here let the critic overwrite the recurrent layers' outputs
        # under the running GRU kernel, which then spun on its hand-off sentinel until the timeout)
        g["keep"] = []
        g["fwd"] = torch.cuda.CUDAGraph()
        g["fake_out"] = _capture(g["fwd"], fwd_body, dev, g["keep"])
        g["critic"] = torch.cuda.CUDAGraph()
        g["critic_out"] = _capture(g["critic"], critic_body, dev, g["keep"])
        g["critic_grads"] = [p.grad for p in self.critic.parameters()]
        g["gen"] = torch.cuda.CUDAGraph()
        g["gen_out"] = _capture(g["gen"], gen_body, dev, g["keep"], pool=g["critic"].pool())
        g["gen_grads"] = [p.grad for p in self.gen.parameters()]
        with torch.no_grad():
            for m, bufs in zip(mods, saved):
                for b, v in zip(m.buffers(), bufs):
                    b.copy_(v)
        torch.cuda.synchronize(dev)
        self._graphs[key] = g
        return g


# =========================================================================================== phase 1
class Phase1Engine(WganGpEngine):
    """Still-pose WGAN-GP (phase1/train_wgan-gp.py:79-110). Dropout stays active in both
    networks during every pass, as in the reference (it never calls .eval())."""

    def __init__(self, gen, critic, cfg, **kw):
        super().__init__(gen, critic, cfg["lr_gen"], cfg["lr_critic"], cfg["n_critic_steps"], **kw)
        self.gamma = float(cfg["gamma"])
        self.latent = int(cfg["latent_vector_size"])
        self.host_noise = True

    def _noise(self, B, device):
        if self.host_noise:
            return host_draw("randn", (B, self.latent), device)
        return torch.randn(B, self.latent, device=device)

    def critic_iteration(self, real):
        out = self._critic_body(real)
        self._begin_critic_step()
        return out

    def _critic_body(self, real):
        B = real.size(0)
        noise = self._noise(B, real.device)
        with torch.no_grad():
            fake = self.gen(noise)
        self._finish_critic_step()
        self.optim_critic.zero_grad(set_to_none=True)  # after the deferred step used the old gradients
        gp = gradient_penalty(self.critic, B, real, fake, device=real.device)
        err_real = self.critic(real).mean()
        err_fake = self.critic(fake).mean()
        err_critic = err_fake - err_real + self.gamma * gp
        err_critic.backward()
        return {"loss_critic": err_critic.detach(), "gp": gp.detach(), "w_dist": (err_fake - err_real).detach()}

    def generator_iteration(self, real):
        self._finish_critic_step()
        out = self._generator_body(real)
        self._gen_step()
        return out

    def _generator_body(self, real):
        B = real.size(0)
        self.optim_gen.zero_grad(set_to_none=True)
        fake = self.gen(self._noise(B, real.device))
        with _Freeze(self.critic):
            # the critic's dropout draws one mask per call, real first (train_wgan-gp.py:100-101)
            with torch.no_grad():
                err_real = self.critic(real).mean()
            err_fake = self.critic(fake).mean()
            err_gen = err_real - err_fake
            err_gen.backward()
        return {"loss_gen": err_gen.detach()}

    # ------------------------------------------------------------------ captured-graph mode
    def enable_graphs(self, on=True):
        """Replay each loop body's forward / backward as one captured HIP graph per batch shape: the phase-1 networks
        are a few dozen sub-10 us launches per pass, so the eager loop is bound by the host's launch rate (2.8 ms per
        body for 0.75 ms of kernels at batch 64). Optimizer steps and the gradient exchange stay eager. Every HOST draw
        of the body - generator noise, interpolation weights, the dropout masks of both networks (the reference never
        calls .eval()) - is recorded on a `layers.DrawTape` and made again, in the same order on the same generator,
        before each replay: results equal the eager path's (tests/test_critic_step.py)."""
        self._use_graphs = bool(on)
        self._graphs = {}
        if on and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        return self

    def train_step(self, real, inputs_ready=None):
        if not getattr(self, "_use_graphs", False) or real.device.type != "cuda":
            return super().train_step(real, inputs_ready=inputs_ready)
        self.total_iterations += 1
        self._finish_critic_step()
        g = self._graph_for(real)
        g["real"].copy_(real.reshape(g["real"].shape))
        g["tape_c"].refill()
        Phase3Engine._bind_grads(self.critic, g["critic_grads"])
        g["critic"].replay()
        self._begin_critic_step()
        out = dict(g["critic_out"])
        if self.total_iterations % self.n_critic_steps == 0:
            self._finish_critic_step()
            g["tape_g"].refill()
            Phase3Engine._bind_grads(self.gen, g["gen_grads"])
            g["gen"].replay()
            self._gen_step()
            out.update(g["gen_out"])
        self.last = out
        self.last_full.update(out)
        return out

    def _graph_for(self, real):
        key = tuple(real.shape)
        g = self._graphs.get(key)
        if g is not None:
            return g
        dev = real.device
        g = {"real": torch.empty_like(real).copy_(real), "tape_c": DrawTape(), "tape_g": DrawTape()}
        K = kernels.impl()

        def critic_body():
            g["tape_c"].rewind()
            with K.weight_cache(), draw_tape(g["tape_c"]):
                return self._critic_body(g["real"])

        def gen_body():
            g["tape_g"].rewind()
            with K.weight_cache(), draw_tape(g["tape_g"]):
                return self._generator_body(g["real"])

        mods = [self.gen, self.critic]
        saved = [[b.clone() for b in m.buffers()] for m in mods]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        quiet = [x.suspended() for x in (self.x_critic, self.x_gen) if x is not None]
        with torch.cuda.stream(side), contextlib.ExitStack() as es:
            for q in quiet:
                es.enter_context(q)
            critic_body()  # (records the tapes; the static buffers hold zeros: values do not matter here)
            gen_body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.optim_critic.zero_grad(set_to_none=True)
        self.optim_gen.zero_grad(set_to_none=True)
        g["keep"] = []
        g["critic"] = torch.cuda.CUDAGraph()
        g["critic_out"] = _capture(g["critic"], critic_body, dev, g["keep"])
        g["critic_grads"] = [p.grad for p in self.critic.parameters()]
        g["gen"] = torch.cuda.CUDAGraph()
        g["gen_out"] = _capture(g["gen"], gen_body, dev, g["keep"], pool=g["critic"].pool())
        g["gen_grads"] = [p.grad for p in self.gen.parameters()]
        with torch.no_grad():
            for m, bufs in zip(mods, saved):
                for b, v in zip(m.buffers(), bufs):
                    b.copy_(v)
        torch.cuda.synchronize(dev)
        self._graphs[key] = g
        return g


# =========================================================================================== synthetic data
def synthetic_phase3_batch(B, T, device, seed=0, audio_rate=16000, video_rate=25, window_s=0.2, lazy=None,
                           with_event=False):
    """Random poses / audio of the dataset's shapes: poses U[0,1) (B, T, 69),
    audio N(0, 0.1^2) (B, T*640), windows of 3200 samples every 640. On a HIP device the windows are
    the in-place view of the padded track (lazy slicing: the generator's first conv gathers them) and the values come
    from the DEVICE generator seeded with `seed` (the train scripts draw one such batch per loop body: the host
    generator needs 35 ms for the 5 M normal draws, three loop bodies' worth of GPU time); on the CPU from the host's.
    with_event: stage the batch on the copy stream (no host sync with the compute stream) and also return
    the event after which it is complete - what `train_step(..., inputs_ready=)` takes."""
    from .utils import slice_audio_batch
    hop = audio_rate // video_rate
    window = int(window_s * audio_rate)
    device = torch.device(device)
    if lazy is None:
        lazy = device.type == "cuda"

    def draw():
        g = torch.Generator(device=device).manual_seed(seed)
        real = torch.rand(B, T, 69, generator=g, device=device)
        audio = 0.1 * torch.randn(B, hop * T, generator=g, device=device)
        return real, audio, slice_audio_batch(audio, window, hop, window - hop, lazy=lazy)

    if not with_event or device.type != "cuda":
        real, audio, slices = draw()
        return (real, audio, slices, None) if with_event else (real, audio, slices)
    cur = torch.cuda.current_stream(device)
    cs = copy_stream(device)
    with torch.cuda.stream(cs):
        real, audio, slices = draw()
        ready = cs.record_event()
    cur.wait_event(ready)
    for t in (real, audio, slices):
        t.record_stream(cur)
    return real, audio, slices, ready
