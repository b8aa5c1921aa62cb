"""Optimizer step for the engines' gradients (the step
either side of fwd+bwd).  Keras semantics of create_optimizer (recman/tf/core/utils.py:201-213):
Adam(beta .9/.999, epsilon 1e-7 outside the sqrt), Adagrad(initial accumulator 0.1,
epsilon 1e-7); plus plain SGD / momentum (the reference names for those do not exist in
tf.optimizers and raise).

Dense parameters are updated with torch foreach ops (plumbing).  Embedding-side
gradients arrive in IndexedSlices form (rows + idx); this version densifies them with
rm_scatter_add_rows - exactly what the reference's dense l2 term forces (layers.py:188-193) -
which is fine for ml-100k-sized tables.  A row-wise sparse update for Criteo-sized tables
is the next item."""
import math

import torch


class Optimizer:
    def __init__(self, name="adam", lr=1e-3):
        if name not in ("adam", "adagrad", "gd", "sgd", "momentum"):
            raise ValueError(f"unknown optimizer {name!r}")  # utils.py:213
        self.name, self.lr = name, float(lr)
        self.t = 0
        self.state = {}

    def reset(self):
        """Forget every moment: the reference builds a NEW optimizer for every batch
        (xDeepFM.py:121-126); strict_reference mode calls this before each step."""
        self.t = 0
        self.state = {}

    @torch.no_grad()
    def step(self, params, grads):
        """params / grads: name -> tensor (same shapes).  In-place update."""
        self.t += 1
        for k, g in grads.items():
            p = params[k]
            if self.name == "adam":
                st = self.state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
                m, v = st
                m.mul_(0.9).add_(g, alpha=0.1)
                v.mul_(0.999).addcmul_(g, g, value=0.001)
                lr_t = self.lr * math.sqrt(1 - 0.999 ** self.t) / (1 - 0.9 ** self.t)
                p.addcdiv_(m, v.sqrt().add_(1e-7), value=-lr_t)
            elif self.name == "adagrad":
                acc = self.state.setdefault(k, torch.full_like(p, 0.1))
                acc.addcmul_(g, g)
                p.addcdiv_(g, acc.sqrt().add_(1e-7), value=-self.lr)
            elif self.name == "momentum":
                buf = self.state.setdefault(k, torch.zeros_like(p))
                buf.mul_(0.9).add_(g)
                p.add_(buf, alpha=-self.lr)
            else:
                p.add_(g, alpha=-self.lr)
