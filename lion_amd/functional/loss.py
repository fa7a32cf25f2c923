"""kl_loss / huber_loss of the PVCNN functional API (third_party/pvcnn/functional/loss.py), pure torch:
   kl_loss(x, y)      = mean_b sum_c KL(softmax(x) || softmax(y)), gradient to y only;
   huber_loss(e, d)   = mean(0.5 e^2 if |e| <= d else d (|e| - 0.5 d))."""
import torch
import torch.nn.functional as F

__all__ = ["kl_loss", "huber_loss"]


def kl_loss(x, y):
    target = F.softmax(x.detach(), dim=1)
    return F.kl_div(F.log_softmax(y, dim=1), target, reduction="none").sum(dim=1).mean()


def huber_loss(error, delta):
    mag = error.abs()
    return torch.where(mag <= delta, 0.5 * mag * mag, delta * (mag - 0.5 * delta)).mean()
