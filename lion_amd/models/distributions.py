"""Diagonal Normal used by the VAE (reference models/distributions.py:17-37)."""
import numpy as np
import torch


class Normal:
    def __init__(self, mu, log_sigma, sigma=None):
        self.mu = mu
        self.log_sigma = log_sigma
        self.sigma = torch.exp(log_sigma) if sigma is None else sigma

    def sample(self, t=1.0):
        rho = torch.randn_like(self.mu)
        return rho * (self.sigma * t) + self.mu, rho

    def sample_given_rho(self, rho):
        return rho * self.sigma + self.mu

    def mean(self):
        return self.mu

    def log_p(self, samples):
        z = (samples - self.mu) / self.sigma
        return -0.5 * z * z - 0.5 * np.log(2 * np.pi) - self.log_sigma
