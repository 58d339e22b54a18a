"""`img_size` / `channels` helpers of the reference (/root/reference/src/downscaling/tf_utils.py:7-12).
`shortcut_convolution` (tf_utils.py:15-32) is unreachable from make_discriminator."""


def img_size(z):
    return z.shape[2]


def channels(z):
    return z.shape[-1]
