"""kl_loss / huber_loss -- mirrors third_party/pvcnn/functional/loss.py (pure torch)."""
import torch
import torch.nn.functional as F

__all__ = ["kl_loss", "huber_loss"]


def kl_loss(x, y):
    p = F.softmax(x.detach(), dim=1)
    log_q = F.log_softmax(y, dim=1)
    return torch.mean(torch.sum(p * (torch.log(p) - log_q), dim=1))


def huber_loss(error, delta):
    abs_error = torch.abs(error)
    quadratic = torch.clamp(abs_error, max=delta)
    return torch.mean(0.5 * quadratic ** 2 + delta * (abs_error - quadratic))
