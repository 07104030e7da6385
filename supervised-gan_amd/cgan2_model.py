"""CGANModel of `--model cgan2` (models/cgan2_model.py:14-298): the conditional GAN trainer with a SECOND, unpaired label
image.  `set_input` takes (real_A, real_B) from input['A'] and fake_A from input['B'] (:122-124); every forward runs the
generator on both labels (:137-138); `--train_D_on_fake_fake_pair` / `--train_G_on_fake_fake_pair` choose whether the
discriminator step / the generator step sees (real_A, G(real_A)) or (fake_A, G(fake_A)) (:169-178, :200-210); the weighted
L1 term exists only on the paired label and `loss_G_L1` is kept unscaled (:219-232).  Everything else -- networks,
discriminator loss, optimizers, checkpoints, LR schedule -- is CGANModel's, on the same MI355X kernels."""
from collections import OrderedDict

import torch

from . import networks
from .cgan_model import CGANModel
from .image_pool import ImagePool


class CGAN2Model(CGANModel):
    def name(self):
        return 'cGANModel'

    def initialize(self, opt):
        assert opt.dataset_mode == 'unaligned'                      # cgan2_model.py:34
        CGANModel.initialize(self, opt)
        self.input_fake_A = self.Tensor(opt.batchSize, opt.input_nc, opt.fineSize, opt.fineSize)
        if self.isTrain:
            self.fake_pool = ImagePool(opt.pool_size, reject=opt.pool_reject_prob)

    def set_input(self, input):
        a = input['A'].to(self.device, non_blocking=True)
        fa = input['B'].to(self.device, non_blocking=True).index_select(1, self._chnl_dev[0])
        ia, ib = a.index_select(1, self._chnl_dev[0]), a.index_select(1, self._chnl_dev[1])
        self.input_A.resize_(ia.size()).copy_(ia)
        self.input_B.resize_(ib.size()).copy_(ib)
        self.input_fake_A.resize_(fa.size()).copy_(fa)
        self.image_paths = input.get('A_paths')

    def forward(self):
        self.real_A = self.input_A
        self.fake_A = self.input_fake_A
        self.real_B = self.input_B
        self.noise = self._draw_noise()
        self.fake_B_from_real_A = self.netG.forward(self.real_A, self.noise)
        self.fake_B_from_fake_A = self.netG.forward(self.fake_A, self.noise)
        self.fake_B = self.fake_B_from_real_A

    sample_noise = forward

    def test(self):
        with torch.no_grad():
            self.forward()

    def _pair(self, fake_fake):
        if fake_fake:
            return self.fake_A, self.fake_B_from_fake_A
        return self.real_A, self.fake_B_from_real_A

    def _pool_source(self):
        a, b = self._pair(self.opt.train_D_on_fake_fake_pair)
        return b if self.opt.no_cgan else networks.cat_pair(a, b)

    def backward_G(self):
        """loss_G = sum_i lambda_i * GAN(D_i(pair), 1) + lambda_A * L1_w(G(real_A), real_B) [paired label only]"""
        from . import networks  # noqa: F401  (criterion classes live there)
        skip = getattr(self.opt, 'skip_wasted_D_wgrad', False)
        for netD in self.netD:
            netD.compute_param_grads = not skip
        a, b = self._pair(self.opt.train_G_on_fake_fake_pair)
        fake = b if self.opt.no_cgan else networks.cat_pair(a, b)
        trick = not self.opt.no_logD_trick
        self.loss_G_GAN, self._each_G = self._d_losses([(d, fake, trick) for d in self.netD],
                                                       [l if trick else -l for l in self.opt.lambda_D])
        for netD in self.netD:
            netD.compute_param_grads = True
        if not self.opt.train_G_on_fake_fake_pair:
            self.loss_G_L1 = self.criterionL1.from_labels(self.fake_B_from_real_A, self.real_B, self.real_A, self.opt.weights, 1.0)
            self.loss_G = self.loss_G_GAN + self.loss_G_L1 * self.opt.lambda_A
        else:
            self.loss_G_L1 = torch.zeros((), device=self.device)
            self.loss_G = self.loss_G_GAN
        self._backward(self.loss_G)

    def get_current_errors(self):
        return OrderedDict([('G_GAN', float(self.loss_G.detach())), ('D_real', float(self.loss_D_real)),
                            ('D_fake', float(self.loss_D_fake))])

    def get_current_visuals(self, save_as_single_image=False):
        if self.isTrain:
            return OrderedDict([('real_A', self.real_A.detach()), ('fake_B_real_A', self.fake_B_from_real_A.detach()),
                                ('fake_A', self.fake_A.detach()), ('fake_B_fake_A', self.fake_B_from_fake_A.detach()),
                                ('real_B', self.real_B.detach())])
        return OrderedDict([('real_A', self.real_A.detach()), ('fake_B', self.fake_B.detach())])
