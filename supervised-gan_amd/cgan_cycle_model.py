"""CGANCycleModel (models/cgan_cycle_model.py:14-290): conditional GAN with an inverse generator.  G1: label A -> image B,
G2: image B -> label A, discriminators D1 on cat(A, B); the generator step updates G1 and G2 together on
    GAN(D1(cat(A, G1(A)))) + lambda_A L1_w(G1(A), B) + lambda_B BCE(G2(B), A) + lambda_A_cycle BCE(G2(G1(A)), A)     (:188-225)
(BCE on the [-1, 1] -> [0, 1] rescaled maps).  Same method names, loss definitions, update order, optimizer groups
(G1 at --lr1, G2 at --lr2) and checkpoint names (G1, G2, D1_n) as the reference, on the MI355X kernels."""
from collections import OrderedDict

import torch

from . import networks, ops
from .base_model import BaseModel
from .image_pool import ImagePool
from .optim import AdamGroups, FusedAdam


class CGANCycleModel(BaseModel):
    allow_multi_G_updates = False

    def name(self):
        return 'CGANCycleModel'

    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        self.isTrain = opt.isTrain
        idx_dict = {'r': 0, 'g': 1, 'b': 2}
        self.chnl_idx_input = [[idx_dict[c] for c in s] for s in opt.which_channel.split('_')]
        assert len(self.chnl_idx_input) == 2
        opt.input_nc = len(self.chnl_idx_input[0])
        opt.output_nc = len(self.chnl_idx_input[1])
        self._chnl_dev = [torch.tensor(ix, dtype=torch.long, device=self.device) for ix in self.chnl_idx_input]
        self.input_A = self.Tensor(opt.batchSize, opt.input_nc, opt.fineSize, opt.fineSize)
        self.input_B = self.Tensor(opt.batchSize, opt.output_nc, opt.fineSize, opt.fineSize)
        self.noise1_ = self.Tensor(opt.batchSize, opt.noise_nc1, opt.noiseSize1, opt.noiseSize1)
        self.noise2_ = self.Tensor(opt.batchSize, opt.noise_nc2, opt.noiseSize2, opt.noiseSize2)
        self.noise1 = self.noise2 = None
        self._rng_seed = 0 if opt.manualSeed is None else int(opt.manualSeed)
        self._rng_offset = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.noise_source = None    # optional callable(which) -> z tensor, which in {1, 2} (tests inject latents)

        self.netG1 = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf1, opt.which_model_netG1, opt.norm, not opt.no_dropout1,
                                       n_layers_G=opt.n_layers_G1, use_residual=False, use_fcn=opt.noiseSize1 != 1,
                                       noise_nc=opt.noise_nc1, add_gaussian_noise=opt.add_gaussian_noise,
                                       gaussian_sigma=opt.gaussian_sigma, upsample_mode=opt.upsample_mode1,
                                       n_layers_CRN_block=opt.n_layers_CRN_block1,
                                       share_label_weights=not opt.no_share_label_block_weights1,
                                       n_layers_G_skip=opt.n_layers_G1_skip, gpu_ids=self.gpu_ids)
        self.netG2 = networks.define_G(opt.output_nc, opt.input_nc, opt.ngf2, opt.which_model_netG2, opt.norm, not opt.no_dropout2,
                                       n_layers_G=opt.n_layers_G2, use_residual=False, use_fcn=opt.noiseSize2 != 1,
                                       noise_nc=opt.noise_nc2, add_gaussian_noise=opt.add_gaussian_noise,
                                       gaussian_sigma=opt.gaussian_sigma, upsample_mode=opt.upsample_mode2,
                                       n_layers_CRN_block=opt.n_layers_CRN_block2,
                                       share_label_weights=not opt.no_share_label_block_weights2,
                                       n_layers_G_skip=opt.n_layers_G2_skip, gpu_ids=self.gpu_ids)
        if self.isTrain:
            assert (len(opt.scale_factor1) == len(opt.lambda_D1) == len(opt.n_layers_D1))
            # the reference's sample_noise (:140-146) does not regenerate fake_A: its second backward_G of a step walks a freed
            # autograd graph and raises
            assert opt.n_update_G == 1 or self.allow_multi_G_updates, \
                "cgan_cycle: --n_update_G > 1 fails in the reference as well (stale fake_A graph)"
            self.n_netD1 = len(opt.scale_factor1)
            self.netD1 = []
            d_nc = opt.output_nc if opt.no_cgan else opt.output_nc + opt.input_nc
            for scale, n_layers in zip(opt.scale_factor1, opt.n_layers_D1):
                d = networks.define_D(d_nc, opt.ndf1, opt.which_model_netD1, n_layers_D=n_layers, norm=opt.norm,
                                      use_sigmoid=opt.no_lsgan1, scale_factor=scale, gpu_ids=self.gpu_ids)
                d.fuse_sigmoid_into_loss = True
                self.netD1.append(d)
            if self.gpu_ids:
                networks.pack_flat(self.netD1)
        if self.isTrain and opt.sequential_train:
            for label, net in (('G1', self.netG1), ('G2', self.netG2)):
                if label in opt.which_model_to_load:
                    self.load_network(net, label, opt.which_epoch_sequential, model_dir=opt.pretrained_model_dir)
            if 'D1' in opt.which_model_to_load:
                for n, netD in enumerate(self.netD1):
                    self.load_network(netD, 'D1_%d' % n, opt.which_epoch_sequential, model_dir=opt.pretrained_model_dir)
        if not self.isTrain or opt.continue_train:
            self.load_network(self.netG1, 'G1', opt.which_epoch)
            self.load_network(self.netG2, 'G2', opt.which_epoch)
            if self.isTrain:
                for n, netD in enumerate(self.netD1):
                    self.load_network(netD, 'D1_%d' % n, opt.which_epoch)
        if self.isTrain:
            self.fake_pool1 = ImagePool(opt.pool_size)
            self.old_lr, self.old_lr1, self.old_lr2 = opt.lr, opt.lr1, opt.lr2
            self.criterionGAN1 = networks.GANLoss(use_lsgan=not opt.no_lsgan1)
            self.criterionL1 = networks.WeightedL1Loss()
            self.optimizer_G = AdamGroups([{'name': 'G1', 'params': self.netG1.parameters(), 'lr': opt.lr1},
                                           {'name': 'G2', 'params': self.netG2.parameters(), 'lr': opt.lr2}],
                                          lr=opt.lr, betas=(opt.beta1, 0.999))
            self.optimizer_D1 = FusedAdam([p for d in self.netD1 for p in d.model.parameters()], lr=opt.lr1, betas=(opt.beta1, 0.999))
            self.grad_sync = None
            self._pool_overrides = None

    # ---- hipGraph hooks (graph_step.GraphedStep) --------------------------------------------------
    def _pair(self, a, b):
        return b if self.opt.no_cgan else networks.cat_pair(a, b)

    def graph_spec(self):
        o = self.opt
        assert (o.n_update_D1, o.n_update_G) == (1, 1), "graphed cgan_cycle step: one update each"
        prog = [[self.optimizer_D1.zero_grad, self.backward_D1], ("sync", self.optimizer_D1),
                [self.optimizer_D1.step, self.optimizer_G.zero_grad, self.backward_G], ("sync", self.optimizer_G),
                [self.optimizer_G.step]]
        return dict(pools=[self.fake_pool1], sources=lambda: [self._d_fake_source()],
                    set_overrides=lambda views: setattr(self, "_pool_overrides", views), program=prog)

    # ---- data ---------------------------------------------------------------------------------
    def set_input(self, input):
        AtoB = self.opt.which_direction == 'AtoB'
        if self.opt.dataset_mode == 'aligned':
            a, b = input['A' if AtoB else 'B'], input['B' if AtoB else 'A']
        elif self.opt.dataset_mode == 'single':
            a = b = input['A']
        else:
            raise NotImplementedError('Dataset mode [%s] is not recognized' % self.opt.dataset_mode)
        a = a.to(self.device, non_blocking=True).index_select(1, self._chnl_dev[0])
        b = b.to(self.device, non_blocking=True).index_select(1, self._chnl_dev[1])
        self.input_A.resize_(a.size()).copy_(a)
        self.input_B.resize_(b.size()).copy_(b)
        self.image_paths = input.get('A_paths' if AtoB else 'B_paths')

    def _draw(self, which):
        buf = self.noise1_ if which == 1 else self.noise2_
        if self.noise_source is not None:
            buf.copy_(self.noise_source(which))
        else:
            ops.normal_fill(buf, self._rng_seed + which, self._rng_offset)
        return buf

    def forward(self):
        """(:129-138)"""
        self.real_A, self.real_B = self.input_A, self.input_B
        self.noise1, self.noise2 = self._draw(1).clone(), self._draw(2).clone()
        self.fake_B = self.netG1.forward(self.real_A, self.noise1)
        self.fake_A = self.netG2.forward(self.real_B, self.noise2)
        self.recon_A = self.netG2.forward(self.fake_B, self.noise2)

    def sample_noise(self):
        """(:140-146): fake_A is NOT regenerated"""
        self.noise1, self.noise2 = self._draw(1).clone(), self._draw(2).clone()
        self.fake_B = self.netG1.forward(self.real_A, self.noise1)
        self.recon_A = self.netG2.forward(self.fake_B, self.noise2)

    def test(self):
        with torch.no_grad():
            self.real_A = self.input_A
            self.noise1, self.noise2 = self._draw(1).clone(), self._draw(2).clone()
            self.fake_B = self.netG1.forward(self.real_A, self.noise1)

    def get_image_paths(self):
        return self.image_paths

    # ---- losses ---------------------------------------------------------------------------------
    def _d_fake_source(self):
        """What backward_D1 hands to ImagePool.query."""
        return self._pair(self.real_A, self.fake_B)

    def _gan(self, jobs, weights):
        preds = networks.multi_forward([(d, x) for d, x, _ in jobs])
        return self.criterionGAN1.weighted_sum(preds, [r for _, _, r in jobs], weights)

    def backward_D1(self):
        """(:162-186)"""
        if self._pool_overrides is not None:
            fake = self._pool_overrides[0]
        else:
            fake = self.fake_pool1.query(self._d_fake_source())
        fake = fake.detach()
        real = self._pair(self.real_A, self.real_B)
        n = self.n_netD1
        self.loss_D, each = self._gan([(d, fake, False) for d in self.netD1] + [(d, real, True) for d in self.netD1], [0.5] * (2 * n))
        self.loss_D_fake, self.loss_D_real = each[:n].sum(), each[n:].sum()
        self._backward(self.loss_D)

    def backward_G(self):
        """(:188-225)"""
        o = self.opt
        for netD in self.netD1:
            netD.compute_param_grads = not getattr(o, 'skip_wasted_D_wgrad', False)
        trick = not o.no_logD_trick
        self.loss_G_GAN, _ = self._gan([(d, self._pair(self.real_A, self.fake_B), trick) for d in self.netD1],
                                       [l if trick else -l for l in o.lambda_D1])
        for netD in self.netD1:
            netD.compute_param_grads = True
        self.loss_G_L1 = self.criterionL1.from_labels(self.fake_B, self.real_B, self.real_A, o.weights, 1.0)
        self.loss_G_CE = networks.bce_on_rescaled(self.fake_A, self.real_A)
        self.loss_G_cycle = networks.bce_on_rescaled(self.recon_A, self.real_A)
        self.loss_G = self.loss_G_GAN + self.loss_G_L1 * o.lambda_A + self.loss_G_CE * o.lambda_B + self.loss_G_cycle * o.lambda_A_cycle
        self._backward(self.loss_G)

    def optimize_parameters(self):
        ops.begin_step()      # one launch zeroes every statistics arena of the step
        o = self.opt
        self.forward()
        for n_up, opt_, back in ((o.n_update_D1, self.optimizer_D1, self.backward_D1), (o.n_update_G, self.optimizer_G, self.backward_G)):
            for _ in range(n_up):
                opt_.zero_grad()
                back()
                if self.grad_sync is not None:
                    self.grad_sync(opt_)
                opt_.step()
                if n_up > 1:
                    self.sample_noise()

    def get_current_errors(self):
        return OrderedDict([('G1', float(self.loss_G.detach())), ('G2', float(self.loss_G_cycle.detach())), ('D1', float(self.loss_D.detach()))])

    def get_current_visuals(self, save_as_single_image=False):
        if self.isTrain:
            return OrderedDict([('real_A', self.real_A.detach()), ('real_B', self.real_B.detach()), ('fake_B', self.fake_B.detach()),
                                ('recon_A', self.recon_A.detach())])
        return OrderedDict([('real_A', self.real_A.detach()), ('fake_B', self.fake_B.detach())])

    def save(self, label):
        self.save_network(self.netG1, 'G1', label, gpu_ids=self.gpu_ids)
        self.save_network(self.netG2, 'G2', label, gpu_ids=self.gpu_ids)
        for n, netD in enumerate(self.netD1):
            self.save_network(netD, 'D1_%d' % n, label, gpu_ids=self.gpu_ids)

    def update_learning_rate(self):
        """(:270-289): G1 / D1 follow lr1, G2 follows lr2."""
        nd = self.opt.niter_decay
        lr = max(0, self.old_lr - self.opt.lr / nd)
        lr1 = max(0, self.old_lr1 - self.opt.lr1 / nd)
        lr2 = max(0, self.old_lr2 - self.opt.lr2 / nd)
        for g in self.optimizer_D1.param_groups:
            g['lr'] = lr1
        for g in self.optimizer_G.param_groups:
            g['lr'] = lr1 if g.get('name') == 'G1' else lr2 if g.get('name') == 'G2' else lr
        self.optimizer_D1.sync_lr()
        self.optimizer_G.sync_lr()
        print('update learning rate: %f -> %f, %f -> %f' % (self.old_lr1, lr1, self.old_lr2, lr2))
        self.old_lr, self.old_lr1, self.old_lr2 = lr, lr1, lr2
