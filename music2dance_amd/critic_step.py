"""The critic iteration as ONE hand-scheduled pass over the HIP kernels (no autograd tape).

The reference's critic iteration (phase3/train.py:204-216, phase2/train.py:146-155) is three critic
forwards (interpolated / real / fake poses) and three autograd passes (the penalty's first backward with
create_graph, its double backward, the loss backward), losses.py:28-44. The critics are piecewise linear
(conv / linear + ReLU, 'id' or 'relu' heads), so every one of those passes is the same linear operator chain
with fixed activation masks, and the whole iteration can be scheduled by hand:

  forward        pose branch ONCE over 3B rows [interpolated | real | fake] (m2d_pose_pack3 writes them
                 channels-first from the loader's (B, T, 69) poses and the generator's rows); audio branch
                 once over B rows (the audio is never interpolated: SURVEY.md A.6); head over 3B rows.
  backward-data  ONE chain for the pose branch: row cotangents 1 (interpolated rows: this IS the penalty's first
                 backward, d score / d input), -1/B (real), +1/B (fake: the loss backward). The skip
                 connections' gradients are added in the conv epilogues (m2d_conv1d_bwd_data_res).
  penalty        per-sample norms of the interpolated rows' input gradient (+ the audio term), loss scalars.
  tangent        the penalty's double backward is the forward-mode tangent G = d pen / d v pushed through the
                 linearised net: conv_fwd(G) with the forward's masks, written IN PLACE over the interpolated
                 rows of the saved activations (nothing reads those rows afterwards).
  weight grads   dW_n = correlate(x~_{n-1}, h_n) over all 3B rows in ONE launch per layer: rows [0, B) pair
                 (tangent, first-backward gradient) - the second-order term -, rows [B, 3B) pair (activation,
                 loss gradient) - the ordinary term. Bias gradients sum over the ordinary rows only
                 (m2d_conv1d_bwd_weight_from). No gradient-accumulation adds, no concatenations, no transposes.

Same arithmetic as the autograd path of losses.gradient_penalty + critic.score_pair (tests compare the two
gradient for gradient); ~100 launches per iteration instead of ~380.

Heads with `activ: tanh` (phase3/archis/default.py:309-310,339-340 of the reference) are not piecewise linear: with the
code e = tanh(u), the penalty's input gradient is v = A^T diag(1 - e^2) c (A the linearised branch up to u, c the
cotangent arriving from the fc1 / fc2 head), so its derivative has one more term than the tangent pairing above:
d pen = [tangent terms, the tangent passing through tanh as t_e = t_u (1 - e^2)] + <w, du>, w = t_u tanh''(u) c =
tanh_bwd(tanh_bwd_bwd(t_u, c, e), e) - an ORDINARY gradient of the interpolated rows with cotangent w at u. Those rows
then need both their tangent and their activations, so the pose branch runs on a 4B-row layout (activation side
[tangent | interpolated | real | fake], gradient side [first backward | tanh'' chain | real | fake]; row r pairs with
row r in the one weight-gradient launch per layer), the audio branch adds w to the cotangent of its forward half.
"""
import os

import torch

from . import kernels, ops

ACT_NONE, ACT_RELU = ops.ACT_NONE, ops.ACT_RELU


def K():
    return kernels.impl()


def _conv_params(conv):
    return conv.weight, conv.bias, conv.stride[0], conv.padding[0]


class CriticStep:
    """critic: phase3 SequenceDiscriminator / AblatedSequenceDiscriminator or phase2 SequenceDiscriminator."""

    @staticmethod
    def supports(critic):
        stick = getattr(critic, "stick_d", critic)
        if not hasattr(stick, "conv1") or not hasattr(stick, "blocks"):
            return False
        heads = [bool(getattr(b, "_head_tanh", False)) for b in (stick, getattr(critic, "audio_d", None)) if b is not None]
        if any(heads) and not (all(heads) and hasattr(critic, "fc1")):
            return False  # (both branches carry the same `activ`; a tanh head always feeds the fc1 / fc2 head)
        return True

    def __init__(self, critic, gamma, lp=False):
        assert self.supports(critic)
        self.critic = critic
        self.gamma = float(gamma)
        self.lp = bool(lp)
        self.stick = getattr(critic, "stick_d", critic)
        self.audio = getattr(critic, "audio_d", None)
        self.has_head = hasattr(critic, "fc1")
        self.fconv = self.stick.fconv if hasattr(self.stick, "fconv") else self.stick.lastconv
        self.head_act = int(getattr(self.stick, "_head_act", ACT_NONE))
        self.tanh = bool(getattr(self.stick, "_head_tanh", False))
        self._const = {}
        self.debug = None  # dev aid: a dict collects clones of the intermediates (tools/critic_step_debug.py)
        # the pose branch's launches are small (they leave most CUs idle between dependent kernels): they run on a
        # side stream underneath the audio branch's large convolutions, section by section
        self.overlap = self.audio is not None and getattr(type(critic), "overlap_branches", True)
        # the penalty's first backward and the score backward of the audio branch as ONE 2B-row launch per layer
        # (M2D_MERGE_AUDIO_BWD=0: two B-row chains, the round-4 schedule)
        self.merge_audio_chains = os.environ.get("M2D_MERGE_AUDIO_BWD", "1") != "0"
        self._side = None
        self.join_pairs = None   # diagnostics (engine.timed_wait): how long the pose branch waited for the generator forward

    # ------------------------------------------------------------------ helpers
    def _constants(self, B, dev, dtype=torch.float32):
        """-> (c1, cw, gamma): c1 (3B, 1) = the score cotangents of [interpolated | real | fake] = (1, -1/B, +1/B);
        cw = the same on the gradient-side row layout (tanh heads: (1, 0, -1/B, +1/B) over 4B rows, else c1)"""
        key = (B, str(dev), dtype)
        c = self._const.get(key)
        if c is None:
            parts = [torch.ones(B), torch.full((B,), -1.0 / B), torch.full((B,), 1.0 / B)]
            c1 = torch.cat(parts).view(3 * B, 1).to(dev, dtype)
            cw = torch.cat([parts[0], torch.zeros(B)] + parts[1:]).view(4 * B, 1).to(dev, dtype) if self.tanh else c1
            c = self._const[key] = (c1, cw, torch.tensor(self.gamma, dtype=dtype).to(dev))
        return c

    def _fork(self, dev):
        """-> (side stream or None, main stream); the side stream starts behind everything queued on main"""
        if not self.overlap or dev.type != "cuda":
            return None, None
        if self._side is None:
            # the critic's own pose-branch stream (its autograd path forks the same way): packed weight images are
            # cached per (weight, stream)
            if getattr(self.critic, "_stick_stream", None) is None:
                self.critic._stick_stream = torch.cuda.Stream(device=dev)
                self.critic._joins = {}
            self._side = self.critic._stick_stream
        cur = torch.cuda.current_stream(dev)
        self._side.wait_stream(cur)
        return self._side, cur

    @staticmethod
    def _join(side, cur, *tensors):
        if side is None:
            return
        cur.wait_stream(side)
        for t in tensors:
            if t is not None:
                t.record_stream(cur)

    class _On:
        def __init__(self, stream):
            self.ctx = torch.cuda.stream(stream) if stream is not None else None

        def __enter__(self):
            if self.ctx is not None:
                self.ctx.__enter__()

        def __exit__(self, *exc):
            if self.ctx is not None:
                return self.ctx.__exit__(*exc)
            return False

    # ------------------------------------------------------------------ the pass
    @torch.no_grad()
    def run(self, real, fake_rows, audio=None, alpha=None, on_grads=None, fake_ready=None):
        """real: (B, T, C) poses [any view of B*T*C], fake_rows: (B*T, C) generator rows (no graph), audio:
        (B, 1, S) for the two-branch critic, alpha: (B, 1) or (B,) interpolation weights on the device.
        Sets p.grad of every critic parameter (they must be None on entry: the engines zero with set_to_none);
        on_grads: called whenever further gradients are in place (GradExchange.poll).
        fake_ready: event after which `fake_rows` is complete (a generator forward still running on its own stream):
        only the pose branch waits for it - the audio branch's forward does not read the poses and starts at once.
        -> {"loss_critic", "gp", "w_dist"} (0-dim device tensors).

        Row layout of the pose branch. Piecewise-linear heads ('id', 'relu'): 3B rows [interpolated | real | fake];
        the penalty's tangent is written IN PLACE over the interpolated rows of the saved activations. `tanh` heads:
        4B rows, activation side [tangent | interpolated | real | fake], gradient side [first backward of the
        interpolated rows | the tanh'' chain | real | fake] - row r of one side pairs with row r of the other in the
        weight-gradient launches (see the module docstring)."""
        k = K()
        st, au = self.stick, self.audio
        dev = fake_rows.device
        dt = fake_rows.dtype
        C = st.conv1.weight.shape[1]
        B = real.size(0)
        T = real.numel() // (B * C)
        tanh = self.tanh
        i0 = B if tanh else 0                 # first forward row
        R = i0 + 3 * B
        fw, itp, tg = slice(i0, R), slice(i0, i0 + B), slice(0, B)   # forward rows, interpolated rows, tangent rows
        c1, cw, gamma_t = self._constants(B, dev, dt)
        dbg = self.debug
        nb = len(st.blocks)
        w1, b1, _, pad1 = _conv_params(st.conv1)
        CH = w1.shape[0]
        Cc = self.fconv.weight.shape[0]
        fw2d = self.fconv.weight.view(Cc, CH * T)

        def rows(ch):
            return torch.empty((R, ch, T), dtype=dt, device=dev)

        # ---------------------------------------------------------------- forward
        side, cur = self._fork(dev)
        if fake_ready is not None:
            first = side if side is not None else torch.cuda.current_stream(dev)
            if self.join_pairs is not None:
                from .engine import timed_wait
                timed_wait(first, fake_ready, self.join_pairs)
            else:
                first.wait_event(fake_ready)
            fake_rows.record_stream(first)
        with self._On(side):
            X = torch.empty((R, C, T), dtype=dt, device=dev)
            k.pose_pack3(real.reshape(B, T, C), fake_rows, alpha.reshape(B), out=X[fw])
            a = [rows(CH)]
            k.conv1d_fwd(X[fw], w1, b1, 1, pad1, ACT_RELU, out=a[0][fw])
            p, q = [], []
            for blk in st.blocks:
                wa, ba, _, pa = _conv_params(blk.conv1)
                wb, bb, _, pb = _conv_params(blk.conv2)
                pk, qk, ak = rows(CH), rows(CH), rows(CH)
                k.conv1d_fwd(a[-1][fw], wa, ba, 1, pa, ACT_RELU, out=pk[fw])
                k.conv1d_fwd(pk[fw], wb, bb, 1, pb, ACT_RELU, residual=a[-1][fw], out=qk[fw], sum_out=ak[fw])
                p.append(pk)
                q.append(qk)
                a.append(ak)
            if tanh:
                # the activation masks of the interpolated rows, once more in the tangent block: the first backward then
                # runs as one launch per layer over all 4B rows with row-aligned masks (the tangent pass overwrites
                # a / p afterwards; q stays - it masks the interpolated rows' gradients in the weight-gradient launches)
                for t_ in a + p + q:
                    t_[tg].copy_(t_[itp])
        Ca = au.l6.weight.shape[0] if au is not None else 0
        E = Cc + Ca
        e = torch.empty((R, E), dtype=dt, device=dev)
        ef = e[fw]
        ea = None
        if au is not None:
            layers = [au.l1, au.l2, au.l3, au.l4, au.l5]
            Y, x = [], audio
            for conv in layers:
                w, b, s_, pd = _conv_params(conv)
                Lo = kernels.conv_out_len(x.shape[2], w.shape[2], s_, pd)
                buf = torch.empty((2 * B, w.shape[0], Lo), dtype=dt, device=dev)
                x = k.conv1d_fwd(x, w, b, s_, pd, ACT_RELU, out=buf[B:])
                Y.append(buf)
            l6w2d = au.l6.weight.view(Ca, -1)
            if tanh:
                ea = k.tanh_fwd(k.gemm_ld(0, Y[-1][B:].view(B, -1), l6w2d, au.l6.bias, ACT_NONE))   # (B, Ca)
                ef.view(3, B, E)[:, :, Cc:] = ea
            else:
                k.gemm_ld(0, Y[-1][B:].view(B, -1), l6w2d, au.l6.bias, self.head_act, out=ef[0:B, Cc:])
                ef.view(3, B, E)[1:, :, Cc:] = ef[0:B, Cc:]
        es = None
        with self._On(side):
            if tanh:
                es = k.tanh_fwd(k.gemm_ld(0, a[-1][fw].view(3 * B, CH * T), fw2d, self.fconv.bias, ACT_NONE))  # (3B, Cc)
                ef[:, :Cc] = es
            else:
                k.gemm_ld(0, a[-1].view(R, CH * T), fw2d, self.fconv.bias, self.head_act, out=e[:, :Cc])
        self._join(side, cur, e)
        if tanh:
            e[tg] = e[itp]   # (gradient-side alignment, as for the masks above; the head's tangent lands here later)
        if dbg is not None:
            dbg.update({"X3": X[fw].clone(), "e": ef.clone(), **{"a%d" % i: t[fw].clone() for i, t in enumerate(a)},
                        **{"p%d" % i: t[fw].clone() for i, t in enumerate(p)}, **{"q%d" % i: t[fw].clone() for i, t in enumerate(q)}})

        # ---------------------------------------------------------------- head forward + first backward
        c_s_raw = c_a_raw = None
        if self.has_head:
            fc1, fc2 = self.critic.fc1, self.critic.fc2
            z = torch.empty((R, fc1.weight.shape[0]), dtype=dt, device=dev)
            k.gemm(0, ef, fc1.weight, fc1.bias, ACT_RELU, out=z[fw])
            s = k.gemm(0, z[fw], fc2.weight, fc2.bias)
            if tanh:
                z[tg] = z[itp]
            dzp = k.gemm(1, cw, fc2.weight, out_mask=z)            # (R, 128), multiplied by relu'(z); tanh: rows [B, 2B) = 0
            de_s = k.gemm_ld(1, dzp, fc1.weight[:, :Cc])           # (R, Cc)
            de_a = k.gemm_ld(1, dzp, fc1.weight[:, Cc:]) if au is not None else None
        else:
            s, dzp, de_s, de_a = e, None, c1, None
        if self.head_act == ACT_RELU:
            de_s = de_s * (e[:, :Cc] > 0)
            if de_a is not None:
                de_a = de_a * (e[:, Cc:] > 0)
        elif tanh:
            # through tanh: cotangent (1 - e^2); the raw cotangents of the interpolated rows are kept for the tanh'' term
            c_s_raw = de_s[tg].contiguous()
            de_s = k.tanh_bwd(de_s, e[:, :Cc].contiguous())
            if de_a is not None:
                c_a_raw = de_a[tg].contiguous()
                de_a = k.tanh_bwd(de_a, e[:, Cc:].contiguous())

        # ---------------------------------------------------------------- backward-data chains
        side, cur = self._fork(dev)
        with self._On(side):
            da = [rows(CH) for _ in range(nb + 1)]
            dp = [rows(CH) for _ in range(nb)]
            k.gemm(1, de_s, fw2d, out=da[nb].view(R, CH * T))
            for i in range(nb - 1, -1, -1):
                blk = st.blocks[i]
                wa, _, _, pa = _conv_params(blk.conv1)
                wb, _, _, pb = _conv_params(blk.conv2)
                k.conv1d_bwd_data(da[i + 1], wb, T, 1, pb, dy_mask=q[i], out_mask=p[i], out=dp[i])
                k.conv1d_bwd_data(dp[i], wa, T, 1, pa, residual=da[i + 1], out_mask=a[0] if i == 0 else None, out=da[i])
            if nb == 0:
                da[0] = da[0] * (a[0] > 0)
            v_pose = k.conv1d_bwd_data(da[0][0:B], w1, T, 1, pad1)
            pen_p, norms_p = k.gp_penalty_fwd(v_pose.view(B, -1), self.lp)
        v_audio = pen_a = None

        def audio_chain(half):
            """backward-data through the audio branch for rows `half` of the (2B, ...) gradient buffers; half = None:
            BOTH halves in one launch per layer - they pass through the same activation masks (the forward's, rows B:
            of Y), which the kernel reads once more for the second half (m2d_conv1d_bwd_data_shared_mask)"""
            for hf in ((lo, hi) if half is None else (half,)):
                k.gemm(1, ca2[hf], l6w2d, out_mask=y5, out=HD[4][hf].view(B, -1))
            for n in range(4, 0, -1):
                w, _, s_, pd = _conv_params(layers[n])
                if half is None:
                    k.conv1d_bwd_data(HD[n], w, Y[n - 1].shape[2], s_, pd, out_mask=Y[n - 1][B:], out=HD[n - 1])
                else:
                    k.conv1d_bwd_data(HD[n][half], w, Y[n - 1].shape[2], s_, pd, out_mask=Y[n - 1][B:], out=HD[n - 1][half])

        if au is not None:
            lo, hi = slice(0, B), slice(B, 2 * B)
            ca2 = torch.empty((2 * B, Ca), dtype=dt, device=dev)
            ca2[lo] = de_a[0:B]
            torch.add(de_a[R - 2 * B:R - B], de_a[R - B:], out=ca2[hi])
            HD = [torch.empty_like(Y[n]) for n in range(5)]
            y5 = Y[4][B:].view(B, -1)
            if tanh or not self.merge_audio_chains:
                audio_chain(lo)
                if not tanh:
                    audio_chain(hi)   # (tanh heads: after the tangent pass - the tanh'' cotangent joins this half)
            else:
                audio_chain(None)
            w, _, s_, pd = _conv_params(layers[0])
            v_audio = k.conv1d_bwd_data(HD[0][lo], w, audio.shape[2], s_, pd)
            pen_a, norms_a = k.gp_penalty_fwd(v_audio.view(B, -1), False)
        self._join(side, cur, pen_p, v_pose)
        if dbg is not None:
            dbg.update({"s": s.clone(), "de_s": de_s.clone(), "v_pose": v_pose.clone(),
                        **{"da%d" % i: t.clone() for i, t in enumerate(da)}, **{"dp%d" % i: t.clone() for i, t in enumerate(dp)}})
            if dzp is not None:
                dbg["dzp"] = dzp.clone()

        # ---------------------------------------------------------------- loss scalars, tangent seeds
        losses = k.wgan_critic_loss(s.view(-1), B, pen_p, pen_a, self.gamma)
        side, cur = self._fork(dev)
        with self._On(side):
            # G = gamma * d pen / d v over the tangent rows of X (in place for 3B layouts: those poses are not read again)
            k.gp_penalty_bwd(v_pose.view(B, -1), norms_p, gamma_t, self.lp, out=X[tg].view(B, -1))
            # ------------------------------------------------------------ tangent through the pose branch
            k.conv1d_fwd(X[tg], w1, None, 1, pad1, ACT_NONE, out_mask=a[0][itp], out=a[0][tg])
            for i, blk in enumerate(st.blocks):
                wa, _, _, pa = _conv_params(blk.conv1)
                wb, _, _, pb = _conv_params(blk.conv2)
                k.conv1d_fwd(a[i][tg], wa, None, 1, pa, ACT_NONE, out_mask=p[i][itp], out=p[i][tg])
                k.conv1d_fwd(p[i][tg], wb, None, 1, pb, ACT_NONE, residual=a[i][tg], out_mask=q[i][itp],
                             out=a[i + 1][tg])
            if self.has_head:
                if tanh:
                    # tangent at u, then (a) through tanh into the head's code rows, (b) the tanh'' cotangent
                    # w = t_u tanh''(u) c = tanh_bwd(tanh_bwd_bwd(t_u, c, e), e), which travels down the branch as an
                    # ordinary gradient of the INTERPOLATED rows (gradient-side rows [B, 2B))
                    t_u = k.gemm_ld(0, a[nb][tg].view(B, CH * T), fw2d)
                    e_i = es[0:B].contiguous()
                    e[tg, :Cc] = k.tanh_bwd(t_u, e_i)
                    de_s[itp] = k.tanh_bwd(k.tanh_bwd_bwd(t_u, c_s_raw, e_i), e_i)
                    k.gemm(1, de_s[itp], fw2d, out=da[nb][itp].view(B, CH * T))
                    for i in range(nb - 1, -1, -1):
                        blk = st.blocks[i]
                        wa, _, _, pa = _conv_params(blk.conv1)
                        wb, _, _, pb = _conv_params(blk.conv2)
                        k.conv1d_bwd_data(da[i + 1][itp], wb, T, 1, pb, dy_mask=q[i][itp], out_mask=p[i][itp], out=dp[i][itp])
                        k.conv1d_bwd_data(dp[i][itp], wa, T, 1, pa, residual=da[i + 1][itp],
                                          out_mask=a[0][itp] if i == 0 else None, out=da[i][itp])
                    if nb == 0:
                        da[0][itp] = da[0][itp] * (a[0][itp] > 0)
                else:
                    hm = e[tg, :Cc] if self.head_act == ACT_RELU else None
                    k.gemm_ld(0, a[nb][tg].view(B, CH * T), fw2d, out_mask=hm, out=e[tg, :Cc])
        if au is not None:
            ga = k.gp_penalty_bwd(v_audio.view(B, -1), norms_a, gamma_t, False).view(audio.shape)
            x = ga
            for n, conv in enumerate(layers):
                w, _, s_, pd = _conv_params(conv)
                x = k.conv1d_fwd(x, w, None, s_, pd, ACT_NONE, out_mask=Y[n][B:], out=Y[n][0:B])
            if tanh:
                t_ua = k.gemm_ld(0, Y[4][0:B].view(B, -1), l6w2d)
                e[tg, Cc:] = k.tanh_bwd(t_ua, ea)
                ca2[hi] += k.tanh_bwd(k.tanh_bwd_bwd(t_ua, c_a_raw, ea), ea)
                audio_chain(hi)
            else:
                hm = e[tg, Cc:] if self.head_act == ACT_RELU else None
                k.gemm_ld(0, Y[4][0:B].view(B, -1), l6w2d, out_mask=hm, out=e[tg, Cc:])
        self._join(side, cur, e)
        if dbg is not None:
            dbg.update({"G0": X[tg].clone(), **{"ga%d" % i: t[tg].clone() for i, t in enumerate(a)},
                        **{"gp%d" % i: t[tg].clone() for i, t in enumerate(p)}})
        if self.has_head:
            # tangent of the head: gz = relu'(z) * (W1 ge), in place over z's tangent rows (they hold the interpolated
            # rows' z: the mask)
            k.gemm(0, e[tg], fc1.weight, out_mask=z[tg], out=z[tg])

        # ---------------------------------------------------------------- weight gradients: one launch per layer
        # in REVERSE parameter order (head, audio branch from its last layer down, pose branch likewise), each p.grad
        # bound as soon as its launch is queued: a data-parallel exchange (`on_grads` = GradExchange.poll) can send a
        # bucket - buckets follow reverse parameter order too - underneath the remaining launches
        late = []  # pose-branch gradients produced on the side stream: bound after the join (a bucket must not be
        #            packed on the main stream before the side stream's launches are ordered in front of it)

        def put(conv, gw, gb, defer=False):
            if defer:
                late.append((conv, gw, gb))
            else:
                conv.weight.grad, conv.bias.grad = gw, gb

        def ready():
            if on_grads is not None:
                on_grads()

        side, cur = self._fork(dev)
        with self._On(side):
            if self.has_head:
                put(self.fconv, k.gemm(2, de_s, a[nb].view(R, CH * T)).view(self.fconv.weight.shape),
                    k.channel_sums(de_s[B:].contiguous()), defer=side is not None)
            else:
                # phase 2: the full-length conv IS the score; its rows pair (tangent, 1) / (activation, +-1/B)
                put(self.fconv, k.gemm(2, c1, a[nb].view(R, CH * T)).view(self.fconv.weight.shape),
                    k.channel_sums(c1[B:].contiguous()), defer=side is not None)
            if side is None:
                ready()
            for i in range(nb - 1, -1, -1):
                blk = st.blocks[i]
                wa, _, _, pa = _conv_params(blk.conv1)
                wb, _, _, pb = _conv_params(blk.conv2)
                put(blk.conv2, *k.conv1d_bwd_weight(p[i], da[i + 1], wb.shape[2], 1, pb, dy_mask=q[i], with_bias=True,
                                                    bias_from_sample=B), defer=side is not None)
                put(blk.conv1, *k.conv1d_bwd_weight(a[i], dp[i], wa.shape[2], 1, pa, with_bias=True,
                                                    bias_from_sample=B), defer=side is not None)
                if side is None:
                    ready()
            put(st.conv1, *k.conv1d_bwd_weight(X, da[0], w1.shape[2], 1, pad1, with_bias=True, bias_from_sample=B),
                defer=side is not None)
        if self.has_head:
            put(fc2, k.gemm(2, cw, z), k.channel_sums(cw[B:].contiguous()))
            put(fc1, k.gemm(2, dzp, e), k.channel_sums(dzp[B:].contiguous()))
        if au is not None:
            put(au.l6, k.gemm(2, ca2, Y[4].view(2 * B, -1)).view(au.l6.weight.shape),
                k.channel_sums(ca2[B:].contiguous()))
            ready()
            for n in range(4, 0, -1):
                w, _, s_, pd = _conv_params(layers[n])
                put(layers[n], *k.conv1d_bwd_weight(Y[n - 1], HD[n], w.shape[2], s_, pd, with_bias=True,
                                                    bias_from_sample=B))
                ready()
            w, _, s_, pd = _conv_params(layers[0])
            g2 = k.conv1d_bwd_weight(ga, HD[0][0:B], w.shape[2], s_, pd)
            g1, gb = k.conv1d_bwd_weight(audio, HD[0][B:], w.shape[2], s_, pd, with_bias=True)
            put(layers[0], g1.add_(g2), gb)
        self._join(side, cur, *[g for _, gw, gb in late for g in (gw, gb)])
        for conv, gw, gb in late:
            conv.weight.grad, conv.bias.grad = gw, gb
        ready()
        return {"loss_critic": losses[0], "gp": losses[1], "w_dist": losses[2]}
