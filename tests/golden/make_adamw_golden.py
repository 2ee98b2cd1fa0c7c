#!/usr/bin/env python3
"""Known-answer vector for ``transformers.optimization.AdamW.step`` as the reference configures it
(trainer_base.py:690-701: AdamW(lr, eps=1e-6), weight decay 0.01 except bias / LayerNorm.weight; transformers==4.2.1,
requirements.txt:1), written INDEPENDENTLY of oracle/vlpet_oracle.py: plain numpy fp64, one scalar formula per line.

transformers 4.2.1 is not installable here (no network; the installed 5.x has no AdamW), so this transcribes the
published update of that release, src/transformers/optimization.py, class AdamW, method step (v4.2.1, lines ~300-352;
https://github.com/huggingface/transformers/blob/v4.2.1/src/transformers/optimization.py):

    for p in group["params"]:
        if p.grad is None: continue                      # a parameter without a gradient is skipped entirely
        state["step"] += 1                               # per-parameter step count
        exp_avg.mul_(beta1).add_(grad, alpha=1.0 - beta1)
        exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1.0 - beta2)
        denom = exp_avg_sq.sqrt().add_(group["eps"])     # eps added to sqrt(v), NOT to sqrt(v / bias_correction2)
        step_size = group["lr"]
        if group["correct_bias"]:                        # default True
            step_size = step_size * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
        p.data.addcdiv_(exp_avg, denom, value=-step_size)
        if group["weight_decay"] > 0.0:                  # decoupled decay AFTER the update, with the plain lr
            p.data.add_(p.data, alpha=-group["lr"] * group["weight_decay"])

Output: tests/golden/adamw_hf421.npz (inputs, per-step gradients with two "grad is None" steps, and p / m / v after
every step for a decayed and an undecayed parameter).  tests/test_oracle_golden.py replays it through the oracle."""
import math
import os

import numpy as np


def adamw_step_scalar(p, g, m, v, step, lr, beta1, beta2, eps, wd):
    """One element, one step; returns (p, m, v)."""
    m = m * beta1 + (1.0 - beta1) * g
    v = v * beta2 + (1.0 - beta2) * g * g
    denom = math.sqrt(v) + eps
    step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p = p - step_size * m / denom
    if wd > 0.0:
        p = p - lr * wd * p
    return p, m, v


def main():
    rng = np.random.RandomState(421)
    n, steps = 257, 6
    beta1, beta2, eps = 0.9, 0.999, 1e-6
    lrs = [0.0, 1e-3, 2e-3, 1.5e-3, 1e-3, 5e-4]                 # a warm-up / decay schedule incl. the lr = 0 first update
    out = dict(hparams=np.array([beta1, beta2, eps]), lrs=np.array(lrs))
    for tag, wd in (("decay", 0.01), ("nodecay", 0.0)):
        p = rng.randn(n)
        p[:8] = [0.0, 1e-8, -1e-8, 1.0, -1.0, 1e3, -1e3, 3.14159]
        m, v = np.zeros(n), np.zeros(n)
        grads = rng.randn(steps, n) * np.array([1.0, 1e-3, 1e-6, 10.0, 0.1, 1e-2])[:, None]
        has_grad = np.array([1, 1, 0, 1, 0, 1])                  # steps 2 and 4: grad is None -> skipped, step count unchanged
        out[f"{tag}::p0"] = p.copy()
        out[f"{tag}::grads"] = grads
        out[f"{tag}::has_grad"] = has_grad
        P, Ms, Vs = [], [], []
        t = 0
        for s in range(steps):
            if has_grad[s]:
                t += 1
                for i in range(n):
                    p[i], m[i], v[i] = adamw_step_scalar(p[i], grads[s, i], m[i], v[i], t, lrs[s], beta1, beta2, eps, wd)
            P.append(p.copy()); Ms.append(m.copy()); Vs.append(v.copy())
        out[f"{tag}::p"], out[f"{tag}::m"], out[f"{tag}::v"] = np.array(P), np.array(Ms), np.array(Vs)
        out[f"{tag}::wd"] = np.array(wd)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "adamw_hf421.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
