"""torch.optim.Adam with its step executed by ONE multi-tensor HIP launch (include/m2d.h: m2d_adam_multi).

The reference builds `torch.optim.Adam(params, lr)` for both networks (phase3/train.py:102-103, phase2/train.py:86-87,
phase1/train_wgan-gp.py:60-61) and calls `.step()` once per iteration. This subclass keeps that object - param_groups,
state_dict layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), `zero_grad`, LR schedulers - and replaces the
arithmetic: every parameter of a group that has a gradient is stepped by the same launch (parameters without one are
skipped and get no state, SURVEY.md A.5: the dead `fc1` / `bn1` affine of every LinearBlock), and conv weights whose
K-major packed images are live in the kernel layer's weight cache have those images rewritten in the same pass (the
critic's weights change every loop body: one launch instead of Adam's three plus a pack launch per conv).
`skip_flag`: optional device float; non-zero at execution time = the step is a no-op on the device."""
import torch

from . import kernels


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False, fused=False)
        self.skip_flag = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        k = kernels.impl()
        skip = self.skip_flag
        if isinstance(skip, int) and skip:
            # the LIVE fault word: a recurrent launch on another stream (the generator forward runs one iteration ahead)
            # may raise it while this step's launches execute, and workgroups that read it before / after would apply /
            # void different tensors. The decision is taken ONCE per step instead: the word is fetched into a device
            # float of this optimizer on the step's stream, and every launch of the step reads that float (data-parallel
            # runs already read a value that was fetched once, into their last gradient bucket).
            dev = next((p.device for g in self.param_groups for p in g["params"] if p.grad is not None), None)
            if dev is not None and dev.type == "cuda":
                if getattr(self, "_skip_buf", None) is None or self._skip_buf.device != dev:
                    self._skip_buf = torch.zeros(1, dtype=torch.float32, device=dev)
                k.fault_fetch(self._skip_buf)
                skip = self._skip_buf
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr = group["lr"]
            lr = float(lr.item()) if torch.is_tensor(lr) else float(lr)
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)   # host tensor, as torch's non-capturable path
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(int(st["step"]), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            # (parameters that got their first gradient later than the others carry their own step count)
            for step, rows in by_step.items():
                ps, gs, ms, vs = zip(*rows)
                k.adam_multi(list(ps), list(gs), list(ms), list(vs), lr, beta1, beta2, group["eps"], step,
                             skip=skip)
        return loss
