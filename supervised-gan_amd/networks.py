"""Drop-in counterpart of the reference's network factory (models/networks.py) on MI355X.

Same entry points -- `define_G`, `define_D`, `GANLoss`, `WeightedL1Loss`, `print_network`,
`weights_init` -- same argument names, same returned-object protocol (`.forward`, `.parameters()`,
`netD.model.parameters()`, `.state_dict()` with the reference's key names and logical shapes,
`netD.gauss_filter`), but a network is a *layer program* over NHWC buffers executed by
hand-written gfx950 kernels (include/sgan_hip.h); normalisation and activations never exist as
separate passes (they are applied while the consumer conv stages its input) and the whole net is
one autograd node.

Implemented: every name of the reference's factories -- which_model_netG in {fcgan, deconv (README alias), fcgan_star, unet_128,
unet_256, crn, autoencoder, dcgan, resnet_6blocks, resnet_9blocks} with their option branches (dropout, batch / instance norm,
residual, Gaussian noise, skips, upsampling modes), which_model_netD in {n_layers, basic, dcgan, n_layers_sep}.  Unknown names
raise NotImplementedError like the reference (models/networks.py:95,123).

This module is the import surface (the factories, `weights_init`, `print_network`); the layer-program core lives in chain.py, the
networks in generators.py / discriminators.py, the losses in losses.py -- re-exported here so that `networks.X` keeps working."""
from __future__ import annotations

import torch

from .ops import pad4      # noqa: F401  (re-export)
from .chain import (BN_EPS, BN_MOMENTUM, IN_EPS, ChainNet, LayerSpec, _BwdArena, _ChainFn, _dgrad_math, _MultiChainFn,      # noqa: F401
                    _ParamBox, can_group, multi_forward, pack_flat)
from .discriminators import (DCGANDiscriminator, NLayerDiscriminator, NLayerDiscriminatorSep, _SigmoidFn, init_gauss_filters,      # noqa: F401
                             matlab_style_gauss2D)
from .generators import (AutoEncoder, CascadedRefinementNetwork, DCGANGenerator, FCGANGenerator, FCGANGeneratorStar,      # noqa: F401
                         ResnetGenerator, UnetGenerator)
from .losses import (GANLoss, GANLossMultiClass, WeightedL1Loss, _CatPairFn, _GanLossFn, _GanLossMultiFn, bce_on_rescaled,      # noqa: F401
                     bilinear_upsample2x, cat_pair)


def weights_init(m):
    """models/networks.py:13-19, applied to our parameter containers (class-name free)."""
    kind = getattr(m, "_sgan_kind", None)
    if kind == "conv":
        m.weight.data.normal_(0.0, 0.02)
        seg = getattr(m.weight, "_sgan_seg", None)
        if seg is not None:      # a write through `.data` moves no version counter: tell the net its derived copies are stale
            seg[0].invalidate_derived()
    elif kind == "bn":
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def print_network(net):
    """models/networks.py:135-140."""
    num_params = sum(p.numel() for p in net.parameters())
    print(net)
    print('Total number of parameters: %d' % num_params)




# ------------------------------------------------------------------------------------------------
# factories (models/networks.py:53-132)
# ------------------------------------------------------------------------------------------------
def define_G(input_nc, output_nc, ngf, which_model_netG, norm='batch', use_dropout=False, n_layers_G=5,
             use_residual=False, use_fcn=False, noise_nc=0, add_gaussian_noise=False, gaussian_sigma=0.1,
             n_layers_G_skip=-1, upsample_mode='convt', share_label_weights=True, n_layers_CRN_block=1, gpu_ids=[]):
    if which_model_netG in ('fcgan', 'deconv'):   # README spells it `deconv` (README.md:33)
        netG = FCGANGenerator(noise_nc, input_nc, ngf, n_layers=n_layers_G, use_dropout=use_dropout, use_fcn=use_fcn,
                              gpu_ids=gpu_ids)
    elif which_model_netG in ('unet_128', 'unet_256'):
        netG = UnetGenerator(input_nc, output_nc, 7 if which_model_netG == 'unet_128' else 8, ngf, norm=norm,
                             use_dropout=use_dropout, use_residual=use_residual, add_gaussian_noise=add_gaussian_noise,
                             gaussian_sigma=gaussian_sigma, num_skips=n_layers_G_skip, gpu_ids=gpu_ids)
    elif which_model_netG == 'crn':
        netG = CascadedRefinementNetwork(input_nc, output_nc, noise_nc, ngf=ngf, n_layers=n_layers_G, norm=norm,
                                         upsample_mode=upsample_mode, add_gaussian_noise=add_gaussian_noise,
                                         gaussian_sigma=gaussian_sigma, share_label_weights=share_label_weights,
                                         n_layers_block=n_layers_CRN_block, gpu_ids=gpu_ids)
    elif which_model_netG == 'autoencoder':
        netG = AutoEncoder(input_nc, output_nc, n_layers_G, ngf, norm=norm, use_dropout=use_dropout, gpu_ids=gpu_ids)
    elif which_model_netG == 'dcgan':
        netG = DCGANGenerator(gpu_ids=gpu_ids, nz=noise_nc, nc=input_nc, ngf=ngf)
    elif which_model_netG == 'fcgan_star':
        netG = FCGANGeneratorStar(noise_nc, input_nc, ngf, n_layers=n_layers_G, use_dropout=use_dropout, use_fcn=use_fcn,
                                  gpu_ids=gpu_ids)
    elif which_model_netG in ('resnet_9blocks', 'resnet_6blocks'):
        netG = ResnetGenerator(input_nc, output_nc, ngf, norm=norm, use_dropout=use_dropout, n_blocks=9 if which_model_netG == 'resnet_9blocks' else 6,
                               use_residual=use_residual, gpu_ids=gpu_ids)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % which_model_netG)
    netG.apply(weights_init)
    if len(gpu_ids) > 0:
        netG.cuda(gpu_ids[0])
    return netG


def define_D(input_nc, ndf, which_model_netD, n_layers_D=3, norm='batch', use_sigmoid=False, scale_factor=1,
             num_classes=2, gpu_ids=[]):
    if which_model_netD == 'basic':
        netD = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm=norm, use_sigmoid=use_sigmoid,
                                   scale_factor=scale_factor, num_classes=num_classes, gpu_ids=gpu_ids)
    elif which_model_netD == 'n_layers':
        netD = NLayerDiscriminator(input_nc, ndf, n_layers=n_layers_D, norm=norm, use_sigmoid=use_sigmoid,
                                   scale_factor=scale_factor, num_classes=num_classes, gpu_ids=gpu_ids)
    elif which_model_netD == 'dcgan':
        if scale_factor > 1:
            raise NotImplementedError("the dcgan discriminator has no Gaussian pre-filter (the reference fails on scale_factor > 1 too)")
        netD = DCGANDiscriminator(gpu_ids=gpu_ids, nc=input_nc, ndf=ndf)
    elif which_model_netD == 'n_layers_sep':
        netD = NLayerDiscriminatorSep(input_nc, ndf, n_layers=n_layers_D, norm=norm, use_sigmoid=use_sigmoid,
                                      scale_factor=scale_factor, num_classes=num_classes, gpu_ids=gpu_ids)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % which_model_netD)
    netD.apply(weights_init)
    if scale_factor > 1:
        for param in netD.gauss_filter.parameters():
            sigma = int(scale_factor) // 2
            kw = 4 * sigma + 1
            param.data = torch.FloatTensor(init_gauss_filters(input_nc, kw, sigma))
    if len(gpu_ids) > 0:
        netD.cuda(gpu_ids[0])
    return netD
