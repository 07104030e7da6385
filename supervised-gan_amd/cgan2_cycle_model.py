"""CGANCycleModel of `--model cgan2_cycle` (models/cgan2_cycle_model.py:14-318): cgan_cycle with a second, unpaired label
image fake_A (from input['B'], :114-121).  Five generator calls per forward (:123-137), `--train_D/G_on_fake_fake_pair`
choose the (label, generated) pair of the discriminator / generator step (:165-176, :202-212), the L1 term exists on the
paired label only (:221-233) and the cycle term splits into a real and a fake part, the latter scaled by
lambda_fake_cycle (:236-246).  Networks, optimizers, checkpoints and LR schedule are CGANCycleModel's."""
from collections import OrderedDict

import torch

from . import networks
from .cgan_cycle_model import CGANCycleModel


class CGAN2CycleModel(CGANCycleModel):
    allow_multi_G_updates = True      # sample_noise regenerates every tensor backward_G reads (:139-149)

    def initialize(self, opt):
        CGANCycleModel.initialize(self, opt)
        self.input_fake_A = self.Tensor(opt.batchSize, opt.input_nc, opt.fineSize, opt.fineSize)

    def set_input(self, input):
        a = input['A'].to(self.device, non_blocking=True)
        fa = input['B'].to(self.device, non_blocking=True).index_select(1, self._chnl_dev[0])
        ia, ib = a.index_select(1, self._chnl_dev[0]), a.index_select(1, self._chnl_dev[1])
        self.input_A.resize_(ia.size()).copy_(ia)
        self.input_B.resize_(ib.size()).copy_(ib)
        self.input_fake_A.resize_(fa.size()).copy_(fa)
        self.image_paths = input.get('A_paths')

    def _generate(self):
        self.noise1, self.noise2 = self._draw(1).clone(), self._draw(2).clone()
        self.fake_B_from_real_A = self.netG1.forward(self.real_A, self.noise1)
        self.fake_B_from_fake_A = self.netG1.forward(self.fake_A, self.noise1)
        self.fake_A_from_real_B = self.netG2.forward(self.real_B, self.noise2)
        self.recon_real_A = self.netG2.forward(self.fake_B_from_real_A, self.noise2)
        self.recon_fake_A = self.netG2.forward(self.fake_B_from_fake_A, self.noise2)
        self.fake_B = self.fake_B_from_real_A

    def forward(self):
        self.real_A, self.real_B, self.fake_A = self.input_A, self.input_B, self.input_fake_A
        self._generate()

    def sample_noise(self):
        self._generate()

    def test(self):
        with torch.no_grad():
            self.real_A = self.input_A
            self.noise1, self.noise2 = self._draw(1).clone(), self._draw(2).clone()
            self.fake_B_from_real_A = self.fake_B = self.netG1.forward(self.real_A, self.noise1)

    def _fake_pair(self, fake_fake):
        if fake_fake:
            return self._pair(self.fake_A, self.fake_B_from_fake_A)
        return self._pair(self.real_A, self.fake_B_from_real_A)

    def _d_fake_source(self):
        return self._fake_pair(self.opt.train_D_on_fake_fake_pair)

    def backward_G(self):
        """(:197-247)"""
        o = self.opt
        for netD in self.netD1:
            netD.compute_param_grads = not getattr(o, 'skip_wasted_D_wgrad', False)
        trick = not o.no_logD_trick
        self.loss_G_GAN, _ = self._gan([(d, self._fake_pair(o.train_G_on_fake_fake_pair), trick) for d in self.netD1],
                                       [l if trick else -l for l in o.lambda_D1])
        for netD in self.netD1:
            netD.compute_param_grads = True
        if not o.train_G_on_fake_fake_pair:
            self.loss_G_L1 = self.criterionL1.from_labels(self.fake_B_from_real_A, self.real_B, self.real_A, o.weights, 1.0)
        else:
            self.loss_G_L1 = torch.zeros((), device=self.device)
        self.loss_G_CE = networks.bce_on_rescaled(self.fake_A_from_real_B, self.real_A)
        self.loss_G_real_cycle = networks.bce_on_rescaled(self.recon_real_A, self.real_A)
        self.loss_G_fake_cycle = networks.bce_on_rescaled(self.recon_fake_A, self.fake_A)
        self.loss_G = self.loss_G_GAN + self.loss_G_L1 * o.lambda_A + self.loss_G_CE * o.lambda_B \
            + self.loss_G_real_cycle * o.lambda_A_cycle + self.loss_G_fake_cycle * o.lambda_A_cycle * o.lambda_fake_cycle
        self._backward(self.loss_G)

    def get_current_errors(self):
        return OrderedDict([('G1', float(self.loss_G.detach())), ('real_cycle', float(self.loss_G_real_cycle.detach())),
                            ('fake_cycle', float(self.loss_G_fake_cycle.detach())), ('D1', float(self.loss_D.detach()))])

    def get_current_visuals(self, save_as_single_image=False):
        if self.isTrain:
            d = lambda t: t.detach()
            return OrderedDict([('real_A', d(self.real_A)), ('fake_B_real_A', d(self.fake_B_from_real_A)), ('fake_A', d(self.fake_A)),
                                ('fake_B_fake_A', d(self.fake_B_from_fake_A)), ('fake_A_real_B', d(self.fake_A_from_real_B)),
                                ('real_B', d(self.real_B)), ('recon_real_A', d(self.recon_real_A)), ('recon_fake_A', d(self.recon_fake_A))])
        return OrderedDict([('real_A', self.real_A.detach()), ('fake_B', self.fake_B_from_real_A.detach())])
