"""TEST-ONLY stand-in for music2dance_amd.kernels.HipKernels on CPU tensors.

It lets the `-m "not gpu"` suite exercise the HOST logic above the kernel layer (autograd
wiring incl. the double backward, module structure, losses, engine, data-parallel exchange)
in a container without a GPU. It is installed explicitly by tests through
`kernels.set_impl(FakeKernels())`; the product never imports this file and has no CPU path.
Each method restates the contract of the corresponding C-ABI entry point (include/m2d.h)
with plain torch ops.
"""
import contextlib

import torch
import torch.nn.functional as F


def _mf(mask, slope):
    return torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, slope))


def _act(y, act, slope):
    if act == 1:
        return F.relu(y)
    if act == 2:
        return F.leaky_relu(y, slope)
    return y


class FakeKernels:
    name = "fake-cpu"

    @contextlib.contextmanager
    def weight_cache(self, keep=False):
        yield self

    def invalidate_packed(self, tensors=None):
        pass

    def adam_multi(self, params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, step, skip=None, repack=True):
        """m2d_adam_multi in torch ops (same arithmetic, element-wise)"""
        if skip is not None and float(skip) != 0.0:
            return
        bc1 = 1.0 - beta1 ** step
        bc2s = (1.0 - beta2 ** step) ** 0.5
        with torch.no_grad():
            for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
                m.add_((g - m) * (1.0 - beta1))
                v.mul_(beta2).add_(g * g * (1.0 - beta2))
                p.sub_((m / (v.sqrt() / bc2s + eps)) * (lr / bc1))

    @staticmethod
    def _into(out, val):
        if out is None:
            return val
        out.copy_(val)
        return out

    def conv1d_fwd(self, x, w, bias, stride, pad, act=0, slope=0.0, residual=None, out_mask=None,
                   out_mask_slope=0.0, with_stats=False, out=None, sum_out=None):
        y = _act(F.conv1d(x, w, bias, stride=stride, padding=pad), act, slope)
        if out_mask is not None:  # mask BEFORE the residual (include/m2d.h)
            y = y * _mf(out_mask, out_mask_slope)
        if sum_out is not None:
            return self._into(out, y), self._into(sum_out, y + residual)
        if residual is not None:
            y = y + residual
        y = self._into(out, y)
        return (y, self.bn_stats(y)) if with_stats else y

    def conv1d_bwd_data(self, dy, w, L, stride, pad, dy_mask=None, dy_mask_slope=0.0, out_mask=None,
                        out_mask_slope=0.0, residual=None, out=None):
        if dy_mask is not None:
            dy = dy * _mf(dy_mask, dy_mask_slope)
        dx = torch.nn.grad.conv1d_input((dy.shape[0], w.shape[1], L), w, dy, stride=stride, padding=pad)
        if residual is not None:  # residual BEFORE the mask
            dx = dx + residual
        if out_mask is not None:
            if out_mask.shape[0] != dx.shape[0]:  # shared mask: the samples behind the mask's read it from its start again
                out_mask = torch.cat([out_mask, out_mask[:dx.shape[0] - out_mask.shape[0]]], 0)
            dx = dx * _mf(out_mask, out_mask_slope)
        return self._into(out, dx)

    def conv1d_bwd_weight(self, x, dy, ks, stride, pad, dy_mask=None, dy_mask_slope=0.0, with_bias=False,
                          bias_from_sample=0):
        if dy_mask is not None:
            dy = dy * _mf(dy_mask, dy_mask_slope)
        dw = torch.nn.grad.conv1d_weight(x, (dy.shape[1], x.shape[1], ks), dy, stride=stride, padding=pad)
        return (dw, dy[bias_from_sample:].sum((0, 2))) if with_bias else dw

    def conv1d_fwd_windows(self, track, T, hop, window, w, bias, stride, pad, act=0, slope=0.0, with_stats=False):
        x = track.unfold(-1, window, hop)[:, :T].reshape(-1, 1, window)
        return self.conv1d_fwd(x, w, bias, stride, pad, act, slope, with_stats=with_stats)

    def conv1d_bwd_weight_windows(self, track, T, hop, window, dy, ks, stride, pad, dy_mask=None, dy_mask_slope=0.0,
                                  with_bias=False):
        x = track.unfold(-1, window, hop)[:, :T].reshape(-1, 1, window)
        return self.conv1d_bwd_weight(x, dy, ks, stride, pad, dy_mask, dy_mask_slope, with_bias)

    def gemm(self, mode, a, b, bias=None, act=0, slope=0.0, a_mask=None, a_mask_slope=0.0, out_mask=None,
             out_mask_slope=0.0, out=None):
        if a_mask is not None:
            a = a * _mf(a_mask, a_mask_slope)
        if mode == 0:
            c = a @ b.t()
        elif mode == 1:
            c = a @ b
        else:
            c = a.t() @ b
        if bias is not None:
            c = c + bias
        c = _act(c, act, slope)
        if out_mask is not None:
            c = c * _mf(out_mask, out_mask_slope)
        return self._into(out, c)

    def gemm_ld(self, mode, a, b, bias=None, act=0, slope=0.0, a_mask=None, a_mask_slope=0.0, out_mask=None,
                out_mask_slope=0.0, out=None):
        return self.gemm(mode, a, b, bias, act, slope, a_mask, a_mask_slope, out_mask, out_mask_slope, out)

    def tanh_fwd(self, x):
        return torch.tanh(x)

    def tanh_bwd(self, gy, y):
        return gy * (1 - y * y)

    def tanh_bwd_bwd(self, g, gy, y):
        return -2 * y * g * gy

    def transposed(self, w):
        return w.t().contiguous()

    def pose_pack3(self, real, fake_rows, alpha, out=None):
        B, T, C = real.shape
        fake = fake_rows.view(B, T, C)
        a = alpha.view(B, 1, 1)
        o = torch.cat((a * real + (1 - a) * fake, real, fake), 0).permute(0, 2, 1).contiguous()
        return self._into(out, o)

    def wgan_critic_loss(self, scores, B, pen0, pen1, gamma):
        s = scores.view(-1)
        gp = pen0 if pen1 is None else pen0 + pen1
        w = s[2 * B:].mean() - s[B:2 * B].mean()
        return torch.stack((w + gamma * gp, gp, w))

    def channel_sums(self, x, mask=None, slope=0.0):
        if mask is not None:
            x = x * _mf(mask, slope)
        return x.sum(dim=(0, 2)) if x.dim() == 3 else x.sum(dim=0)

    @staticmethod
    def _into(out, y):
        """the `out` contract of the kernel layer: write the result into a channel block of a wider buffer"""
        if out is None:
            return y
        out.copy_(y)
        return out

    def bn_fwd(self, x, gamma, beta, running_mean, running_var, training, eps, momentum, act=0, slope=0.0,
               residual=None, out=None):
        dims = (0,) if x.dim() == 2 else (0, 2)
        shape = (1, -1) if x.dim() == 2 else (1, -1, 1)
        if training:
            mean = x.mean(dims)
            var = ((x - mean.view(shape)) ** 2).mean(dims)
            n = x.numel() // x.shape[1]
            if running_mean is not None:
                running_mean.mul_(1 - momentum).add_(momentum * mean)
                running_var.mul_(1 - momentum).add_(momentum * var * (n / (n - 1) if n > 1 else 1.0))
        else:
            mean, var = running_mean.clone(), running_var.clone()
        invstd = 1.0 / torch.sqrt(var + eps)
        y = _act((x - mean.view(shape)) * invstd.view(shape) * gamma.view(shape) + beta.view(shape), act, slope)
        if residual is not None:
            y = y + residual
        return self._into(out, y), mean, invstd

    @staticmethod
    def _interleave(a, b):
        return torch.stack((a.double(), b.double()), 1).reshape(-1)

    def bn_stats(self, x):
        dims = (0,) if x.dim() == 2 else (0, 2)
        xd = x.double()
        return self._interleave(xd.sum(dims), (xd * xd).sum(dims))

    def bn_fwd_sums(self, x, sums, count, gamma, beta, running_mean, running_var, eps, momentum, act=0, slope=0.0,
                    residual=None, out=None):
        shape = (1, -1) if x.dim() == 2 else (1, -1, 1)
        mean64 = sums[0::2] / count
        var64 = (sums[1::2] / count - mean64 * mean64).clamp_min(0.0)
        mean, var = mean64.float(), var64.float()
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * mean)
            running_var.mul_(1 - momentum).add_(momentum * var * (count / (count - 1) if count > 1 else 1.0))
        invstd = (1.0 / torch.sqrt(var64 + eps)).float()
        y = _act((x - mean.view(shape)) * invstd.view(shape) * gamma.view(shape) + beta.view(shape), act, slope)
        if residual is not None:
            y = y + residual
        return self._into(out, y), mean, invstd

    def bn_update_running(self, sums, count, running_mean, running_var, eps, momentum):
        mean64 = sums[0::2] / count
        var64 = (sums[1::2] / count - mean64 * mean64).clamp_min(0.0)
        running_mean.mul_(1 - momentum).add_(momentum * mean64.float())
        running_var.mul_(1 - momentum).add_(momentum * var64.float() * (count / (count - 1) if count > 1 else 1.0))

    def bn_fwd_sums_pool(self, x, sums, count, gamma, beta, running_mean, running_var, eps, momentum, act=0, slope=0.0,
                         out=None):
        y, mean, invstd = self.bn_fwd_sums(x, sums, count, gamma, beta, running_mean, running_var, eps, momentum, act, slope,
                                           None, out)
        return y, F.max_pool1d(y, 2, 2), mean, invstd

    def bn_fwd_sums_upsample2(self, x, sums, count, gamma, beta, running_mean, running_var, eps, momentum, act=0, slope=0.0,
                              out=None):
        y, mean, invstd = self.bn_fwd_sums(x, sums, count, gamma, beta, running_mean, running_var, eps, momentum, act, slope)
        return self._into(out, F.interpolate(y, scale_factor=2, mode="linear", align_corners=False)), mean, invstd

    def _bn_dz(self, dy, x, gamma, beta, save_mean, save_invstd, act, slope):
        shape = (1, -1) if x.dim() == 2 else (1, -1, 1)
        xh = (x - save_mean.view(shape)) * save_invstd.view(shape)
        z = gamma.view(shape) * xh + beta.view(shape)
        dz = dy
        if act == 1:
            dz = dy * (z > 0).to(dy.dtype)
        elif act == 2:
            dz = dy * _mf(z, slope)
        return dz, xh

    def bn_bwd_stats(self, dy, x, gamma, beta, save_mean, save_invstd, act=0, slope=0.0):
        dims = (0,) if x.dim() == 2 else (0, 2)
        dz, xh = self._bn_dz(dy, x, gamma, beta, save_mean, save_invstd, act, slope)
        return self._interleave(dz.double().sum(dims), (dz.double() * xh.double()).sum(dims))

    def bn_bwd_sums(self, dy, x, gamma, beta, save_mean, save_invstd, sums_local, sums_global, count, act=0, slope=0.0):
        shape = (1, -1) if x.dim() == 2 else (1, -1, 1)
        dz, xh = self._bn_dz(dy, x, gamma, beta, save_mean, save_invstd, act, slope)
        m1 = (sums_global[0::2] / count).float()
        m2 = (sums_global[1::2] / count).float()
        dx = gamma.view(shape) * save_invstd.view(shape) * (dz - m1.view(shape) - xh * m2.view(shape))
        return dx, sums_local[1::2].float(), sums_local[0::2].float()

    def bn_bwd(self, dy, x, gamma, beta, save_mean, save_invstd, act=0, slope=0.0):
        dims = (0,) if x.dim() == 2 else (0, 2)
        shape = (1, -1) if x.dim() == 2 else (1, -1, 1)
        xh = (x - save_mean.view(shape)) * save_invstd.view(shape)
        z = gamma.view(shape) * xh + beta.view(shape)
        dz = dy
        if act == 1:
            dz = dy * (z > 0).to(dy.dtype)
        elif act == 2:
            dz = dy * _mf(z, slope)
        n = x.numel() // x.shape[1]
        s1, s2 = dz.sum(dims), (dz * xh).sum(dims)
        dx = gamma.view(shape) * save_invstd.view(shape) * (dz - (s1 / n).view(shape) - xh * (s2 / n).view(shape))
        return dx, s2, s1

    def gru_layer_fwd(self, gi, w_hh_t, b_hh, lengths=None, save=True):
        B, T, H3 = gi.shape
        H = H3 // 3
        h = gi.new_zeros(B, H)
        outs, rs, zs, ns, hns = [], [], [], [], []
        for t in range(T):
            gh = h @ w_hh_t + b_hh
            r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            if lengths is not None:
                h = h * (t < lengths).to(h.dtype).view(B, 1)
            outs.append(h)
            rs.append(r), zs.append(z), ns.append(n), hns.append(gh[:, 2 * H:])
        out = torch.stack(outs, 1)
        saved = torch.stack([torch.stack(v, 1) for v in (rs, zs, ns, hns)], 0) if save else None
        return out, saved

    def gru_layer_bwd(self, dout, out, saved, w_hh, lengths=None):
        B, T, H = out.shape
        r_s, z_s, n_s, hn_s = saved[0], saved[1], saved[2], saved[3]
        dgi = out.new_zeros(B, T, 3 * H)
        dgh = out.new_zeros(B, T, 3 * H)
        dh_next = None
        for t in range(T - 1, -1, -1):
            dh = dout[:, t].clone()
            if dh_next is not None:
                dh = dh + dh_next * z_s[:, t + 1] + dgh[:, t + 1] @ w_hh
            if lengths is not None:
                dh = dh * (t < lengths).to(dh.dtype).view(B, 1)
            r, z, n, hn = r_s[:, t], z_s[:, t], n_s[:, t], hn_s[:, t]
            hprev = out[:, t - 1] if t > 0 else torch.zeros_like(n)
            dn = dh * (1 - z) * (1 - n * n)
            dz = dh * (hprev - n) * z * (1 - z)
            dr = dn * hn * r * (1 - r)
            dgi[:, t] = torch.cat((dr, dz, dn), 1)
            dgh[:, t] = torch.cat((dr, dz, dn * r), 1)
            dh_next = dh
        return dgi, dgh

    def gru_stack_fwd(self, gi0, w_ih_t, b_ih, w_hh_t, b_hh, lengths=None, save=True):
        outs, saved = [], []
        for l in range(len(w_hh_t)):
            gi = gi0 if l == 0 else outs[-1] @ w_ih_t[l] + b_ih[l]
            o, s = self.gru_layer_fwd(gi, w_hh_t[l], b_hh[l], lengths, save)
            outs.append(o), saved.append(s)
        return outs, (saved if save else None)

    def gru_stack_bwd(self, dout, outs, saved, w_hh, w_ih, lengths=None, persistent=True):
        L = len(outs)
        dgi, dgh = [None] * L, [None] * L
        d = dout
        for l in range(L - 1, -1, -1):
            dgi[l], dgh[l] = self.gru_layer_bwd(d, outs[l], saved[l], w_hh[l], lengths)
            if l > 0:
                d = dgi[l] @ w_ih[l]
        return dgi, dgh

    def gp_interpolate(self, real, fake, alpha):
        a = alpha.view(-1, 1)
        return a * real + (1 - a) * fake

    def gp_penalty_fwd(self, g, lp):
        if lp:
            norms = torch.sqrt((g * g).sum(1))
            d = (norms - 1).clamp_min(0)
        else:
            norms = torch.sqrt((g * g).sum(1) + 1e-12)
            d = norms - 1
        return (d * d).mean(), norms

    def gp_penalty_bwd(self, g, norms, gout, lp, out=None):
        d = norms - 1
        if lp:
            d = d.clamp_min(0)
        coef = torch.where(d == 0, torch.zeros_like(d), gout * 2 * d / (norms * g.shape[0]))
        return self._into(out, coef.view(-1, 1) * g)

    def l1_mean_fwd(self, a, b):
        return (a - b).abs().mean()

    def l1_mean_bwd(self, a, b, gout):
        return gout * torch.sign(a - b) / a.numel()

    def tv_mean_fwd(self, x, B, C, T, sb, sc, st):
        v = torch.as_strided(x, (B, C, T), (sb, sc, st))
        return (v[:, :, 1:] - v[:, :, :-1]).abs().mean()

    def tv_mean_bwd(self, x, gout, B, C, T, sb, sc, st):
        v = torch.as_strided(x, (B, C, T), (sb, sc, st))
        s = torch.sign(v[:, :, 1:] - v[:, :, :-1]) * (gout / (B * C * (T - 1)))
        dx = torch.zeros_like(x)
        dv = torch.as_strided(dx, (B, C, T), (sb, sc, st))
        dv[:, :, 1:] += s
        dv[:, :, :-1] -= s
        return dx

    def jerk_mean_fwd(self, x, B, C, T, sb, sc, st):
        v = torch.as_strided(x, (B, C, T), (sb, sc, st))
        d = v[:, :, 3:] - 3 * v[:, :, 2:-1] + 3 * v[:, :, 1:-2] - v[:, :, :-3]
        return (d ** 2).sum(dim=1).mean()

    def affine_cols(self, x, scale, shift, out=None):
        y = x * scale + shift
        if out is not None:
            out.copy_(y)
            return out
        return y

    def maxpool2_fwd(self, x):
        return F.max_pool1d(x, 2, 2)

    def maxpool2_bwd(self, x, dy):
        xr = x.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            y = F.max_pool1d(xr, 2, 2)
        return torch.autograd.grad(y, xr, dy)[0]

    def upsample2_fwd(self, x, out=None):
        return self._into(out, F.interpolate(x, scale_factor=2, mode="linear", align_corners=False))

    def upsample2_bwd(self, dy):
        B, C, Lo = dy.shape
        xr = torch.zeros(B, C, Lo // 2, dtype=dy.dtype, requires_grad=True)
        with torch.enable_grad():
            y = F.interpolate(xr, scale_factor=2, mode="linear", align_corners=False)
        return torch.autograd.grad(y, xr, dy)[0]

    def prof_begin(self):
        pass

    def prof_end(self):
        return {}
