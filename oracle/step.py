"""Oracle (test infrastructure): loss and optimiser step of the reference train loop.

Restates /root/reference/chexpert.py:159-165 (`train_epoch` body) and :530 (loss), with the
optimiser wiring of :470 (Adam), :479-480 (SGD momentum .9 nesterov + MultiStepLR[40000,60000]),
:499-500 (RMSprop momentum .9 eps 1e-3 + ExponentialLR).  torch.optim is the third-party
arithmetic the reference itself calls, so it is used here as-is on CPU tensors.
"""
import torch
import torch.nn.functional as F


def bce_sum_mean(logits, target):
    """BCEWithLogitsLoss(reduction='none')(out, target).sum(1).mean(0)  (chexpert.py:160, :530)."""
    return F.binary_cross_entropy_with_logits(logits, target, reduction="none").sum(1).mean(0)


def make_optimizer(kind, params, lr):
    if kind == "adam":
        return torch.optim.Adam(params, lr=lr), None
    if kind == "sgd_nesterov":
        opt = torch.optim.SGD(params, lr=lr, momentum=0.9, nesterov=True)
        return opt, torch.optim.lr_scheduler.MultiStepLR(opt, [40000, 60000])
    if kind == "rmsprop":
        opt = torch.optim.RMSprop(params, lr=lr, momentum=0.9, eps=0.001)
        return opt, torch.optim.lr_scheduler.ExponentialLR(opt, 0.97)
    raise ValueError(kind)


PARAM_SUFFIXES_EXCLUDED = ("running_mean", "running_var", "num_batches_tracked")


def trainable(sd):
    return [k for k in sd if not k.endswith(PARAM_SUFFIXES_EXCLUDED)]


def train_step(forward, sd, x, target, optimizer=None):
    """One reference minibatch: forward, loss, zero_grad, backward, (optimizer.step()).
    `forward(sd, x)` is one of oracle.nets.*_forward bound to its config.  Returns (loss, logits, grads)."""
    names = trainable(sd)
    for k in names:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    logits = forward(sd, x)
    loss = bce_sum_mean(logits, target)
    loss.backward()
    grads = {k: sd[k].grad.detach().clone() for k in names}
    if optimizer is not None:
        optimizer.step()
    return loss.detach(), logits.detach(), grads
