"""supervised-gan_amd: MI355X-native conv G/D training hot path of phymhan/supervised-gan.

Python host code (this package) keeps the reference's `define_G` / `define_D` / `GANLoss` option
surface and checkpoint layout (models/networks.py:53-132,152-214; models/base_model.py:44-61) and
calls hand-written gfx950 kernels through the C ABI in include/sgan_hip.h (libsgan_hip.so)."""
__version__ = "0.1.0"
