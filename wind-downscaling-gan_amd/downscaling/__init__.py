"""MI355X-native `downscaling` package (drop-in for the GAN hot path of
OpheliaMiralles/wind-downscaling-gan: make_generator / make_discriminator / GAN / FlexibleNoiseGenerator
and the `downscale` driver).  Importing it needs no GPU; running any op does (there is no CPU path)."""
from .api import *  # noqa: F401,F403
