"""CPU oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional, state_dict-driven restatement (plain torch CPU ops, fp32 or fp64) of the
reference's WGAN-GP hot path: every function takes a flat `dict name -> tensor` with the
reference's own state_dict key names and restates one reference function,
cited as file:line into clementabary/music2dance.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module; music2dance_amd/ never does (its ops fail loudly without the HIP library).

Pinning: the reference holds no tests or golden vectors of its own, so
the oracle is pinned against outputs of the reference itself, generated in the build
container by tests/golden/make_golden.py (which imports /root/reference) and committed as
tests/golden/*.npz; tests/test_oracle_golden.py checks every function here against them.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- primitives
def activation(x, kind):
    """'id' | 'relu' | 'tanh' switch used by every encoder / critic head
    (phase3/archis/default.py:72-76,98-103,130-135,305-310,335-340)."""
    if kind == "id":
        return x
    if kind == "relu":
        return F.relu(x)
    if kind == "tanh":
        return torch.tanh(x)
    raise ValueError(kind)


def batch_norm(sd, prefix, x, training):
    """nn.BatchNorm1d(eps=1e-5, momentum=0.1) on (N, C) or (N, C, L); updates the running
    buffers in `sd` when training (phase3/archis/default.py:154,179-180)."""
    w, b = sd[prefix + "weight"], sd[prefix + "bias"]
    dims = (0,) if x.dim() == 2 else (0, 2)
    shape = (1, -1) if x.dim() == 2 else (1, -1, 1)
    if training:
        mean = x.mean(dims)
        var = ((x - mean.view(shape)) ** 2).mean(dims)
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            unbiased = var * (n / (n - 1)) if n > 1 else var
            sd[prefix + "running_mean"] = (1 - BN_MOMENTUM) * sd[prefix + "running_mean"] + BN_MOMENTUM * mean.detach()
            sd[prefix + "running_var"] = (1 - BN_MOMENTUM) * sd[prefix + "running_var"] + BN_MOMENTUM * unbiased.detach()
            sd[prefix + "num_batches_tracked"] = sd[prefix + "num_batches_tracked"] + 1
    else:
        mean, var = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS) * w.view(shape) + b.view(shape)


def linear(sd, prefix, x):
    return x @ sd[prefix + "weight"].t() + sd[prefix + "bias"]


def conv(sd, prefix, x, stride=1, pad=0):
    return F.conv1d(x, sd[prefix + "weight"], sd[prefix + "bias"], stride=stride, padding=pad)


def gru(sd, prefix, x, n_layers, lengths=None):
    """nn.GRU(batch_first=True), gate order (r, z, n), h0 = 0; returns the top layer's
    outputs (NoiseGen.forward, phase3/archis/default.py:349-355). With `lengths`
    (pack_padded_sequence semantics) outputs past a sequence's length are zero."""
    B, T, _ = x.shape
    inp = x
    for layer in range(n_layers):
        w_ih, w_hh = sd["%sweight_ih_l%d" % (prefix, layer)], sd["%sweight_hh_l%d" % (prefix, layer)]
        b_ih, b_hh = sd["%sbias_ih_l%d" % (prefix, layer)], sd["%sbias_hh_l%d" % (prefix, layer)]
        H = w_hh.shape[1]
        gi_all = inp @ w_ih.t() + b_ih
        h = x.new_zeros(B, H)
        outs = []
        for t in range(T):
            gi = gi_all[:, t]
            gh = h @ w_hh.t() + b_hh
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            outs.append(h)
        inp = torch.stack(outs, 1)
        if lengths is not None:
            mask = (torch.arange(T).view(1, T) < torch.as_tensor(lengths).view(B, 1)).to(inp.dtype)
            inp = inp * mask.unsqueeze(-1)
    return inp


def linear_block(sd, prefix, x, use_bn, training):
    """LinearBlock quirk (phase3/archis/default.py:183-192, phase1/archis/residual.py:63-71):
    the fc1 -> bn1 -> relu result is discarded; output = x + relu(bn2(fc2(x))). bn1's running
    statistics still advance in training mode."""
    dead = linear(sd, prefix + "fc1.", x)
    if use_bn:
        batch_norm(sd, prefix + "bn1.", dead.detach(), training)
    y = linear(sd, prefix + "fc2.", x)
    if use_bn:
        y = batch_norm(sd, prefix + "bn2.", y, training)
    return x + F.relu(y)


def frame_decoder(sd, prefix, x, nblocks, training):
    """FrameDecoder.forward (phase3/archis/default.py:163-168, phase2/archis/default.py:116-121)."""
    x = F.relu(batch_norm(sd, prefix + "bn1.", linear(sd, prefix + "fc1.", x), training))
    for i in range(nblocks):
        x = linear_block(sd, "%sblocks.%d." % (prefix, i), x, True, training)
    return linear(sd, prefix + "lastfc.", x)


def temporal_block(sd, prefix, x):
    """TemporalBlock.forward: two k=7 'same' convs with ReLU and a residual add
    (phase3/archis/default.py:207-210)."""
    y = F.relu(conv(sd, prefix + "conv1.", x, 1, 3))
    y = F.relu(conv(sd, prefix + "conv2.", y, 1, 3))
    return x + y


# --------------------------------------------------------------------------- phase 1
def p1_generator(sd, z, nblocks, training, dropout_mask=None):
    """phase1 Generator.forward (phase1/archis/residual.py:21-25). `dropout_mask` is the
    Bernoulli(0.5) keep-mask (already 0/1); None means eval mode / no dropout."""
    x = F.relu(batch_norm(sd, "bn1.", linear(sd, "fc1.", z), training))
    for i in range(nblocks):
        x = linear_block(sd, "blocks.%d." % i, x, True, training)
    if dropout_mask is not None:
        x = x * dropout_mask * 2.0
    return linear(sd, "lastfc.", x)


def p1_critic(sd, x, nblocks, dropout_mask=None):
    """phase1 Discriminator.forward (phase1/archis/residual.py:43-47)."""
    x = F.relu(linear(sd, "fc1.", x.reshape(x.shape[0], -1)))
    for i in range(nblocks):
        x = linear_block(sd, "blocks.%d." % i, x, False, False)
    if dropout_mask is not None:
        x = x * dropout_mask * 2.0
    return linear(sd, "lastfc.", x)


# --------------------------------------------------------------------------- phase 2
def p2_generator(sd, noise, n_cells, n_blocks, training, lengths=None):
    """phase2 SequenceGenerator.forward (phase2/archis/default.py:18-24)."""
    h = gru(sd, "noise_gen.rnn.", noise, n_cells, lengths)
    return frame_decoder(sd, "decoder.", h.reshape(-1, h.shape[-1]), n_blocks, training)


def p2_critic(sd, x, n_blocks, init_ker):
    """phase2 SequenceDiscriminator.forward (phase2/archis/default.py:43-49)."""
    x = F.relu(conv(sd, "conv1.", x, 1, (init_ker - 1) // 2))
    for i in range(n_blocks):
        x = temporal_block(sd, "blocks.%d." % i, x)
    return conv(sd, "lastconv.", x).squeeze(1)


# --------------------------------------------------------------------------- phase 3
def p3_default_encoder(sd, prefix, x, activ, training):
    """DefaultAudioEncoder.forward (phase3/archis/default.py:78-82)."""
    x = conv(sd, prefix + "conv_layers.0.", x, 50, 124)
    x = F.relu(batch_norm(sd, prefix + "activations.0.0.", x, training))
    for i in range(1, 6):
        x = conv(sd, "%sconv_layers.%d." % (prefix, i), x, 2, 1)
        x = F.relu(batch_norm(sd, "%sactivations.%d.0." % (prefix, i), x, training))
    x = activation(conv(sd, prefix + "conv_layers.6.", x), activ)
    return x.squeeze()


def _unet_convblock(sd, prefix, x, training):
    """BasisConvBlock.forward (phase3/archis/default.py:220-221)."""
    return F.leaky_relu(batch_norm(sd, prefix + "bn.", conv(sd, prefix + "conv.", x, 1, 1), training), 0.2)


def p3_unet_encoder(sd, prefix, x, activ, training):
    """UNetAudioEncoder.forward + UBlock.forward (phase3/archis/default.py:105-111,238-246)."""
    x = conv(sd, prefix + "conv_layers.0.", x, 4, 79)
    x = F.leaky_relu(batch_norm(sd, prefix + "activations.0.0.", x, training), 0.2)
    for i in (1, 2):
        x = conv(sd, "%sconv_layers.%d." % (prefix, i), x, 2, 1)
        x = F.leaky_relu(batch_norm(sd, "%sactivations.%d.0." % (prefix, i), x, training), 0.2)
    u = prefix + "ublock."
    up = lambda t: F.interpolate(t, scale_factor=2, mode="linear", align_corners=False)
    x1 = _unet_convblock(sd, u + "convblock1.", x, training)
    x2 = _unet_convblock(sd, u + "convblock2.", F.max_pool1d(x1, 2, 2), training)
    x3 = _unet_convblock(sd, u + "convblock3.", F.max_pool1d(x2, 2, 2), training)
    x4 = _unet_convblock(sd, u + "convblock4.", F.max_pool1d(x3, 2, 2), training)
    x3 = _unet_convblock(sd, u + "convblock5.", torch.cat((up(x4), x3), 1), training)
    x2 = _unet_convblock(sd, u + "convblock6.", torch.cat((up(x3), x2), 1), training)
    x = _unet_convblock(sd, u + "convblock7.", torch.cat((up(x2), x1), 1), training)
    x = activation(conv(sd, prefix + "fc.", x), activ)
    return x.squeeze()


def p3_wavegan_encoder(sd, prefix, x, activ, training):
    """WaveGANAudioEncoder.forward (phase3/archis/default.py:137-143)."""
    for i in (1, 2, 3, 4):
        x = conv(sd, "%sl%d." % (prefix, i), x, 4, 0)
        x = F.relu(batch_norm(sd, "%sbn%d." % (prefix, i), x, training))
    return activation(conv(sd, prefix + "l5.", x), activ).squeeze(-1)


def p3_generator(sd, audio_slices, noise, enc_type, activ, n_cells, n_blocks, training, lengths=None):
    """phase3 SequenceGenerator.forward (phase3/archis/default.py:25-42). `noise` is the
    (B, T, noise_size) tensor the reference draws from the CPU RNG at :31-34."""
    B, T, W = audio_slices.shape
    x = audio_slices.reshape(-1, 1, W)
    enc = {"default": p3_default_encoder, "unet": p3_unet_encoder, "wavegan": p3_wavegan_encoder}[enc_type]
    x = enc(sd, "audio_enc.model.", x, activ, training).view(B, T, -1)
    h = gru(sd, "audio_rnn.rnn.", x, n_cells, lengths)
    n = gru(sd, "noise_gen.rnn.", noise, 1)
    if lengths is not None:
        n = n[:, :h.shape[1]]
    lat = torch.cat((h, n), -1)
    return frame_decoder(sd, "decoder.", lat.reshape(-1, lat.shape[-1]), n_blocks, training)


def p3_stick_critic(sd, prefix, x, init_ker, activ, n_blocks=2):
    """StickDiscriminator.forward (phase3/archis/default.py:342-346)."""
    x = F.relu(conv(sd, prefix + "conv1.", x, 1, (init_ker - 1) // 2))
    for i in range(n_blocks):
        x = temporal_block(sd, "%sblocks.%d." % (prefix, i), x)
    return activation(conv(sd, prefix + "fconv.", x), activ).squeeze(-1)


def p3_audio_critic(sd, prefix, c, activ):
    """AudioDiscriminator.forward (phase3/archis/default.py:312-319)."""
    for i in range(1, 6):
        c = F.relu(conv(sd, "%sl%d." % (prefix, i), c, 4, 11))
    return activation(conv(sd, prefix + "l6.", c), activ).squeeze(-1)


def p3_critic(sd, x, audio, init_ker, activ, ablated=False):
    """SequenceDiscriminator.forward / AblatedSequenceDiscriminator.forward
    (phase3/archis/default.py:263-270,286-291). The ablated variant ignores init_ker and
    uses StickDiscriminator's default 9 (:277-278)."""
    if ablated:
        code = p3_stick_critic(sd, "stick_d.", x, 9, activ)
    else:
        code = torch.cat((p3_stick_critic(sd, "stick_d.", x, init_ker, activ),
                          p3_audio_critic(sd, "audio_d.", audio, activ)), -1)
    return linear(sd, "fc2.", F.relu(linear(sd, "fc1.", code)))


# --------------------------------------------------------------------------- losses
def gradient_penalty(critic_fn, real, fake, alpha, audio=None, is_seq=False, lp=False):
    """losses.gradient_penalty (losses.py:5-60) with the per-sample `alpha` (B, 1) passed in
    (the reference draws it with torch.rand on the CPU RNG, :15). `critic_fn(x)` or
    `critic_fn(x, audio)`. Returns (penalty, pose_term, audio_term_or_None)."""
    B = real.shape[0]
    r = real.reshape(B, -1)
    f = fake.reshape(B, -1)
    a = alpha.expand(r.shape)
    interp = a * r.detach() + (1 - a) * f.detach()
    interp = interp.view(B, 69, -1) if is_seq else interp.view(B, 23, 3)
    interp.requires_grad_(True)
    if audio is not None:
        if not audio.requires_grad:
            audio.requires_grad_(True)
        out = critic_fn(interp, audio)
        inputs = (interp, audio)
    else:
        out = critic_fn(interp)
        inputs = (interp,)
    grads = torch.autograd.grad(out, inputs, torch.ones_like(out), create_graph=True, retain_graph=True)
    g0 = grads[0].reshape(B, -1)
    if audio is None:
        if lp:
            d = g0.norm(2, dim=1) - 1
            d = torch.where(d < 0, torch.zeros_like(d), d)
            t = (d ** 2).mean()
        else:
            t = ((torch.sqrt((g0 ** 2).sum(1) + 1e-12) - 1) ** 2).mean()
        return t, t, None
    g1 = grads[1].reshape(B, -1)
    t0 = ((torch.sqrt((g0 ** 2).sum(1) + 1e-12) - 1) ** 2).mean()
    t1 = ((torch.sqrt((g1 ** 2).sum(1) + 1e-12) - 1) ** 2).mean()
    return t0 + t1, t0, t1


def tv_loss(seq):
    """losses.tv_loss (losses.py:76-82): seq (B, C, T)."""
    return (seq[:, :, 1:] - seq[:, :, :-1]).abs().mean()


def slice_audio(audio, window, hop, pad_samples):
    """utils.slice_audio_batch (utils.py:329-353) == zero-pad pad_samples//2 left, the rest
    right, then all windows of `window` samples every `hop`."""
    left = pad_samples // 2
    return F.pad(audio, (left, pad_samples - left)).unfold(-1, window, hop)


# --------------------------------------------------------------------------- optimiser
class Adam:
    """torch.optim.Adam defaults (betas 0.9/0.999, eps 1e-8, no weight decay) as
    phase3/train.py:102-103 builds it; parameters whose grad is None are skipped and get no
    state, like the dead fc1 / bn1 affine parameters."""

    def __init__(self, lr):
        self.lr = lr
        self.state = {}

    def step(self, params, grads):
        out = {}
        for k, p in params.items():
            g = grads.get(k)
            if g is None:
                out[k] = p
                continue
            st = self.state.setdefault(k, {"t": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
            st["t"] += 1
            st["m"] = 0.9 * st["m"] + 0.1 * g
            st["v"] = 0.999 * st["v"] + 0.001 * g * g
            bc1 = 1 - 0.9 ** st["t"]
            bc2 = 1 - 0.999 ** st["t"]
            denom = st["v"].sqrt() / math.sqrt(bc2) + 1e-8
            out[k] = p - (self.lr / bc1) * st["m"] / denom
        return out


# --------------------------------------------------------------------------- helpers for loops
def is_param(name):
    return not (name.endswith("running_mean") or name.endswith("running_var") or name.endswith("num_batches_tracked"))


def split_state(sd):
    """-> (trainable leaf copies with requires_grad, buffers)"""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if is_param(k)}
    buffers = {k: v.detach().clone() for k, v in sd.items() if not is_param(k)}
    return params, buffers


def grads_of(loss, params, retain_graph=False):
    names = list(params)
    gs = torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True, retain_graph=retain_graph)
    return dict(zip(names, gs))


class P3Config:
    """The ctor arguments phase3/train.py:87-98 passes, with phase3/configs/default.yaml values."""

    def __init__(self, enc_type="default", activ="id", ablated=False, n_cells=3, n_blocks=2, init_ker=25,
                 gamma=10.0, beta=1.0, eta=0.0, lr=2e-4, n_critic=8, T=120, noise_size=10):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def p3_train_iterations(gen_sd, critic_sd, cfg, real, audio, audio_slices, n_iters, rng_seed):
    """Re-enactment of the phase-3 inner loop (phase3/train.py:186-243) on ONE fixed batch:
    `real` (B, T, 69), `audio` (B, samples), `audio_slices` (B, T, window). Noise and alpha
    come from the default CPU generator in the reference's draw order, seeded with
    `rng_seed`. Returns (trace dict of python floats per iteration, final gen_sd, critic_sd)."""
    torch.manual_seed(rng_seed)
    B, T = real.shape[0], real.shape[1]
    g_params, g_buf = split_state(gen_sd)
    d_params, _ = split_state(critic_sd)
    opt_d, opt_g = Adam(cfg.lr), Adam(cfg.lr)
    trace = {"loss_critic": [], "gp": [], "w_dist": [], "loss_gen": [], "err_l1": [], "g_step": []}
    audio_c = audio.unsqueeze(1)
    real_c = real.view(B, T, 69).permute(0, 2, 1).contiguous()

    def gen_forward():
        sd = dict(g_params)
        sd.update(g_buf)
        noise = torch.randn(B, T, cfg.noise_size)
        out = p3_generator(sd, audio_slices, noise, cfg.enc_type, cfg.activ, cfg.n_cells, cfg.n_blocks, True)
        for k in g_buf:
            g_buf[k] = sd[k]
        return out

    def critic(x, a=None):
        return p3_critic(d_params, x, a, cfg.init_ker, cfg.activ, cfg.ablated)

    for it in range(1, n_iters + 1):
        fake = gen_forward().view(B, T, 69).permute(0, 2, 1).contiguous()
        alpha = torch.rand(B, 1)
        a_in = None if cfg.ablated else audio_c.detach().clone()
        gp, _, _ = gradient_penalty(critic, real_c, fake, alpha, a_in, is_seq=True, lp=False)
        a_in2 = None if cfg.ablated else audio_c
        err_real = critic(real_c, a_in2).mean()
        err_fake = critic(fake.detach(), a_in2).mean()
        err_critic = err_fake - err_real + cfg.gamma * gp
        grads = grads_of(err_critic, d_params)
        trace["loss_critic"].append(err_critic.item())
        trace["gp"].append(gp.item())
        trace["w_dist"].append((err_fake - err_real).item())
        new = opt_d.step({k: v.detach() for k, v in d_params.items()}, grads)
        d_params = {k: v.detach().clone().requires_grad_(True) for k, v in new.items()}
        if it % cfg.n_critic:
            trace["g_step"].append(0)
            continue
        trace["g_step"].append(1)
        fake = gen_forward().view(B, T, 69).permute(0, 2, 1)
        err_l1 = (real_c - fake).abs().mean()
        err_real = critic(real_c, a_in2).mean()
        err_fake = critic(fake, a_in2).mean()
        err_gen = err_real - err_fake + cfg.beta * err_l1 + cfg.eta * tv_loss(fake)
        grads = grads_of(err_gen, g_params)
        trace["loss_gen"].append(err_gen.item())
        trace["err_l1"].append(err_l1.item())
        new = opt_g.step({k: v.detach() for k, v in g_params.items()}, grads)
        g_params = {k: v.detach().clone().requires_grad_(True) for k, v in new.items()}
    gen_out = {k: v.detach() for k, v in g_params.items()}
    gen_out.update(g_buf)
    return trace, gen_out, {k: v.detach() for k, v in d_params.items()}


def p2_train_iterations(gen_sd, critic_sd, real, n_iters, rng_seed, n_cells=3, n_blocks_gen=2, n_blocks_critic=3,
                        init_ker=25, gamma=10.0, eta=50.0, lr=5e-4, n_critic=8, noise_size=50, T=120):
    """Re-enactment of the phase-2 `wgangp` loop (phase2/train.py:135-180): WGAN-LP penalty,
    TV-regularised generator loss, MultiStepLR stepped only on generator iterations (the
    milestones 10k/35k/50k are never reached in a short trace, so lr stays constant)."""
    torch.manual_seed(rng_seed)
    B = real.shape[0]
    g_params, g_buf = split_state(gen_sd)
    d_params, _ = split_state(critic_sd)
    opt_d, opt_g = Adam(lr), Adam(lr)
    trace = {"loss_critic": [], "gp": [], "w_dist": [], "loss_gen": [], "g_step": []}
    real_c = real.view(B, T, 69).permute(0, 2, 1).contiguous()

    def gen_forward():
        sd = dict(g_params)
        sd.update(g_buf)
        noise = torch.randn(B, T, noise_size)
        out = p2_generator(sd, noise, n_cells, n_blocks_gen, True)
        for k in g_buf:
            g_buf[k] = sd[k]
        return out

    critic = lambda x: p2_critic(d_params, x, n_blocks_critic, init_ker)
    for it in range(1, n_iters + 1):
        fake = gen_forward().view(B, T, 69).permute(0, 2, 1).contiguous()
        alpha = torch.rand(B, 1)
        gp, _, _ = gradient_penalty(critic, real_c, fake, alpha, None, is_seq=True, lp=True)
        err_real, err_fake = critic(real_c).mean(), critic(fake.detach()).mean()
        err_critic = err_fake - err_real + gamma * gp
        grads = grads_of(err_critic, d_params)
        trace["loss_critic"].append(err_critic.item())
        trace["gp"].append(gp.item())
        trace["w_dist"].append((err_fake - err_real).item())
        new = opt_d.step({k: v.detach() for k, v in d_params.items()}, grads)
        d_params = {k: v.detach().clone().requires_grad_(True) for k, v in new.items()}
        if it % n_critic:
            trace["g_step"].append(0)
            continue
        trace["g_step"].append(1)
        fake = gen_forward().view(B, T, 69).permute(0, 2, 1)
        err_gen = critic(real_c).mean() - critic(fake).mean() + eta * tv_loss(fake)
        grads = grads_of(err_gen, g_params)
        trace["loss_gen"].append(err_gen.item())
        new = opt_g.step({k: v.detach() for k, v in g_params.items()}, grads)
        g_params = {k: v.detach().clone().requires_grad_(True) for k, v in new.items()}
    gen_out = {k: v.detach() for k, v in g_params.items()}
    gen_out.update(g_buf)
    return trace, gen_out, {k: v.detach() for k, v in d_params.items()}


def _keep_mask(shape):
    """The Bernoulli(0.5) keep-mask F.dropout draws from the default CPU generator."""
    return torch.empty(shape).bernoulli_(0.5)


def p1_critic_iteration_values(gen_sd, critic_sd, z, real, rng_seed, nblocks=1, size=128, gamma=10.0):
    """One phase-1 critic iteration (phase1/train_wgan-gp.py:84-92) in train mode with the
    dropout masks and alpha drawn in the reference's order from the CPU RNG."""
    torch.manual_seed(rng_seed)
    B = real.shape[0]
    g_params, g_buf = split_state(gen_sd)
    d_params, _ = split_state(critic_sd)
    sd = dict(g_params)
    sd.update(g_buf)
    fake = p1_generator(sd, z, nblocks, True, _keep_mask((B, size)))
    alpha = torch.rand(B, 1)
    mask_gp = _keep_mask((B, size))
    gp, _, _ = gradient_penalty(lambda x: p1_critic(d_params, x, nblocks, mask_gp), real, fake, alpha)
    err_real = p1_critic(d_params, real, nblocks, _keep_mask((B, size))).mean()
    err_fake = p1_critic(d_params, fake.detach(), nblocks, _keep_mask((B, size))).mean()
    err_critic = err_fake - err_real + gamma * gp
    grads = grads_of(err_critic, d_params)
    return {"fake": fake.detach(), "gp": gp.item(), "err_real": err_real.item(), "err_fake": err_fake.item(),
            "grads": grads, "gen_buffers": {k: sd[k] for k in g_buf}}


def p1_train_iterations(gen_sd, critic_sd, real, n_iters, rng_seed, nblocks=1, latent=10, size=128, gamma=10.0,
                        lr=1e-4, n_critic=5):
    """Re-enactment of phase1/train_wgan-gp.py:79-110 on one fixed batch of real poses."""
    torch.manual_seed(rng_seed)
    B = real.shape[0]
    g_params, g_buf = split_state(gen_sd)
    d_params, _ = split_state(critic_sd)
    opt_d, opt_g = Adam(lr), Adam(lr)
    trace = {"loss_critic": [], "loss_gen": []}

    def gen_forward():
        sd = dict(g_params)
        sd.update(g_buf)
        noise = torch.randn(B, latent)
        out = p1_generator(sd, noise, nblocks, True, _keep_mask((B, size)))
        for k in g_buf:
            g_buf[k] = sd[k]
        return out

    for it in range(1, n_iters + 1):
        fake = gen_forward()
        alpha = torch.rand(B, 1)
        mask_gp = _keep_mask((B, size))
        gp, _, _ = gradient_penalty(lambda x: p1_critic(d_params, x, nblocks, mask_gp), real, fake, alpha)
        err_real = p1_critic(d_params, real, nblocks, _keep_mask((B, size))).mean()
        err_fake = p1_critic(d_params, fake.detach(), nblocks, _keep_mask((B, size))).mean()
        err_critic = err_fake - err_real + gamma * gp
        trace["loss_critic"].append(err_critic.item())
        new = opt_d.step({k: v.detach() for k, v in d_params.items()}, grads_of(err_critic, d_params))
        d_params = {k: v.detach().clone().requires_grad_(True) for k, v in new.items()}
        if it % n_critic:
            continue
        fake = gen_forward()
        err_real = p1_critic(d_params, real, nblocks, _keep_mask((B, size))).mean()
        err_fake = p1_critic(d_params, fake, nblocks, _keep_mask((B, size))).mean()
        err_gen = err_real - err_fake
        trace["loss_gen"].append(err_gen.item())
        new = opt_g.step({k: v.detach() for k, v in g_params.items()}, grads_of(err_gen, g_params))
        g_params = {k: v.detach().clone().requires_grad_(True) for k, v in new.items()}
    gen_out = {k: v.detach() for k, v in g_params.items()}
    gen_out.update(g_buf)
    return trace, gen_out, {k: v.detach() for k, v in d_params.items()}


# ------------------------------------------------------------------------------------------- evaluation metric
def jerkiness(sequence):
    """losses.py:85-89: squared third finite difference along time of (B, C, T), summed over channels,
    mean over (B, T - 3)."""
    d = sequence[:, :, 3:] - 3 * sequence[:, :, 2:-1] + 3 * sequence[:, :, 1:-2] - sequence[:, :, :-3]
    return (d ** 2).sum(dim=1).mean()
