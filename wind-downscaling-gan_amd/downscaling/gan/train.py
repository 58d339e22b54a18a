"""Losses and optimizer factories with the reference names
(/root/reference/src/downscaling/gan/train.py:7-58).  Scores are torch tensors."""
from typing import Callable

import torch

from downscaling.engine.trainer import AdamTF
from downscaling.gan.metrics import wind_speed_weighted_rmse, discriminator_score_real, discriminator_score_fake  # noqa: F401

generator_losses = [wind_speed_weighted_rmse]
scaling_factors = [1.]


def discriminator_loss(real_output, fake_output):
    return -(torch.mean(real_output) - torch.mean(fake_output))


def discriminator_adversarial_loss(real_output, fake_output):
    return -(torch.mean(real_output) - torch.mean(fake_output))


class reconstruction_loss:
    def __init__(self, feature_extractor: Callable[[torch.Tensor], torch.Tensor], coefficient: float = 1):
        self.feature_extractor = feature_extractor
        self.coefficient = coefficient

    def __call__(self, low_res, high_res):
        delta = self.feature_extractor(low_res) - self.feature_extractor(high_res)
        return self.coefficient * torch.mean(torch.sqrt(torch.sum(delta ** 2, dim=-1)))


def generator_loss(real_output, fake_output):
    return torch.mean(torch.stack([sf * loss(real_output, fake_output).mean() for loss, sf in
                                   zip(generator_losses, scaling_factors)]))


def generator_optimizer():
    return AdamTF(lr=1e-4, beta_1=0.5, beta_2=0.9, epsilon=0.1)


def discriminator_optimizer():
    return AdamTF(lr=4e-4, beta_1=0.5, beta_2=0.9, epsilon=0.1)
