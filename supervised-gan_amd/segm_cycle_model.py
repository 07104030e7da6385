"""SegmentationCycleModel (models/segm_cycle_model.py:15-394): G1 image -> class logits (softmax / `--use_sigmoid_ss` sigmoid gives
fake_B), G2 one-hot label -> image, run on the real label (fake_A, judged by the discriminators D2 on cat(label, image)) and on G1's
prediction (recon_A, the cycle); generator loss = lambda_A CE(fake_B) + GAN(D2(fake_A)) + lambda_B L1(fake_B, real_B) +
lambda_A_cycle L1(recon_A, real_A) with G1 stepped at lr1, G2 and D2 at lr2.

Built on CGANCycleModel's plumbing (channel picking, latents, AdamGroups, checkpoints); the segmentation-specific pieces come from
SegmentationModel's restatement (one-hot label, cross-entropy)."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import networks, ops
from .base_model import BaseModel
from .cgan_cycle_model import CGANCycleModel
from .image_pool import ImagePool
from .optim import AdamGroups, FusedAdam
from .losses import softmax_channels
from .segm_model import SegmentationModel, _identity


class SegmentationCycleModel(CGANCycleModel):
    def name(self):
        return 'SegmentationCycleModel'

    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        self.isTrain = opt.isTrain
        idx = {'r': 0, 'g': 1, 'b': 2}
        self.chnl_idx_input = [[idx[c] for c in part] for part in opt.which_channel.split('_')]
        assert len(self.chnl_idx_input) == 2
        opt.input_nc, self.label_nc = len(self.chnl_idx_input[0]), len(self.chnl_idx_input[1])
        self.num_classes = opt.output_nc = self.label_nc + 1 if opt.add_background_onehot else self.label_nc      # :41-42
        self._chnl_dev = [torch.tensor(ix, dtype=torch.long, device=self.device) for ix in self.chnl_idx_input]
        self.class_weights = None if opt.weights is None else torch.tensor(opt.weights, dtype=torch.float32, device=self.device)
        self.use_sigmoid_ss = opt.use_sigmoid_ss
        self.input_A = self.Tensor(opt.batchSize, opt.input_nc, opt.fineSize, opt.fineSize)
        self.input_B = self.Tensor(opt.batchSize, self.num_classes, opt.fineSize, opt.fineSize)
        self.noise1_ = self.Tensor(opt.batchSize, opt.noise_nc1, opt.noiseSize1, opt.noiseSize1)
        self.noise2_ = self.Tensor(opt.batchSize, opt.noise_nc2, opt.noiseSize2, opt.noiseSize2)
        self.noise1 = self.noise2 = self.label = None
        self._rng_seed = 0 if opt.manualSeed is None else int(opt.manualSeed)
        self._rng_offset = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.noise_source = None

        def G(cin, cout, k):
            g = lambda name: getattr(opt, name + k)      # noqa: E731
            return networks.define_G(cin, cout, g('ngf'), g('which_model_netG'), opt.norm, not g('no_dropout'), n_layers_G=g('n_layers_G'),
                                     use_residual=False, use_fcn=g('noiseSize') != 1, noise_nc=g('noise_nc'),
                                     add_gaussian_noise=opt.add_gaussian_noise, gaussian_sigma=opt.gaussian_sigma,
                                     upsample_mode=g('upsample_mode'), n_layers_CRN_block=g('n_layers_CRN_block'),
                                     share_label_weights=not g('no_share_label_block_weights'),
                                     n_layers_G_skip=getattr(opt, 'n_layers_G%s_skip' % k), gpu_ids=self.gpu_ids)
        self.netG1 = G(opt.input_nc, self.num_classes, '1')             # :63-69
        self.netG2 = G(self.num_classes, opt.input_nc, '2')             # :70-76
        if self.isTrain:
            assert (len(opt.scale_factor2) == len(opt.lambda_D2) == len(opt.n_layers_D2))
            self.n_netD2 = len(opt.scale_factor2)
            d_nc = opt.input_nc if opt.no_cgan else opt.input_nc + self.num_classes
            self.netD2 = []
            for scale, n_layers in zip(opt.scale_factor2, opt.n_layers_D2):
                d = networks.define_D(d_nc, opt.ndf2, opt.which_model_netD2, n_layers_D=n_layers, norm=opt.norm, use_sigmoid=opt.no_lsgan2,
                                      scale_factor=scale, gpu_ids=self.gpu_ids)
                d.fuse_sigmoid_into_loss = True
                self.netD2.append(d)
            if self.gpu_ids:
                networks.pack_flat(self.netD2)
        if self.isTrain and opt.sequential_train and not opt.continue_train:
            for label, net in (('G1', self.netG1), ('G2', self.netG2)):
                if label in opt.which_model_to_load:
                    self.load_network(net, label, opt.which_epoch_sequential, model_dir=opt.pretrained_model_dir)
            if 'D2' in opt.which_model_to_load:
                for n, netD in enumerate(self.netD2):
                    self.load_network(netD, 'D2_%d' % n, opt.which_epoch_sequential, model_dir=opt.pretrained_model_dir)
        if not self.isTrain or opt.continue_train:
            self.load_network(self.netG1, 'G1', opt.which_epoch)
            self.load_network(self.netG2, 'G2', opt.which_epoch)
            if self.isTrain:
                for n, netD in enumerate(self.netD2):
                    self.load_network(netD, 'D2_%d' % n, opt.which_epoch)
        if self.isTrain:
            self.fake_pool2 = ImagePool(opt.pool_size)
            self.old_lr, self.old_lr1, self.old_lr2 = opt.lr, opt.lr1, opt.lr2
            self.criterionGAN1 = self.criterionGAN2 = networks.GANLoss(use_lsgan=not opt.no_lsgan2)      # _gan() reads criterionGAN1
            self.criterionL1 = networks.WeightedL1Loss()
            self.optimizer_G = AdamGroups([{'name': 'G1', 'params': self.netG1.parameters(), 'lr': opt.lr1},
                                           {'name': 'G2', 'params': self.netG2.parameters(), 'lr': opt.lr2}],
                                          lr=opt.lr, betas=(opt.beta1, 0.999))
            self.optimizer_D2 = FusedAdam([p for d in self.netD2 for p in d.model.parameters()], lr=opt.lr2, betas=(opt.beta1, 0.999))
            self.grad_sync = None
            self._pool_overrides = None
        SegmentationModel.reset_accs(self)

    def graph_spec(self):
        o = self.opt
        assert (o.n_update_D2, o.n_update_G) == (1, 1), "graphed segmentation_cycle step: one update each"
        prog = [[self.optimizer_D2.zero_grad, self.backward_D2], ("sync", self.optimizer_D2),
                [self.optimizer_D2.step, self.optimizer_G.zero_grad, self.backward_G], ("sync", self.optimizer_G), [self.optimizer_G.step]]
        return dict(pools=[self.fake_pool2], sources=lambda: [self._d_fake_source()],
                    set_overrides=lambda views: setattr(self, "_pool_overrides", views), program=prog)

    # ---- data / forward ---------------------------------------------------------------------------
    def set_input(self, input):
        CGANCycleModel.set_input(self, input)
        SegmentationModel._one_hot_label(self)

    def forward(self):
        """(:159-174)"""
        self.real_A, self.real_B = self.input_A, self.input_B
        self.noise1, self.noise2 = self._draw(1).clone(), self._draw(2).clone()
        self.logit = self.netG1.forward(self.real_A, self.noise1, activation=_identity)
        self.fake_B = torch.sigmoid(self.logit) if self.use_sigmoid_ss else softmax_channels(self.logit)
        self.fake_A = self.netG2.forward(self.real_B, self.noise2)
        self.recon_A = self.netG2.forward(self.fake_B, self.noise2)

    sample_noise = forward          # (:176-185) regenerates everything, fake_A included

    def test(self):
        with torch.no_grad():
            self.real_A, self.real_B = self.input_A, self.input_B
            self.noise1 = self._draw(1).clone()
            self.logit = self.netG1.forward(self.real_A, self.noise1, activation=_identity)
            self.fake_B = torch.sigmoid(self.logit) if self.use_sigmoid_ss else softmax_channels(self.logit)

    # ---- losses -----------------------------------------------------------------------------------
    def _d_fake_source(self):
        return self.fake_A if self.opt.no_cgan else networks.cat_pair(self.real_B, self.fake_A)

    def backward_D2(self):
        """(:201-222)"""
        fake = self._pool_overrides[0] if self._pool_overrides is not None else self.fake_pool2.query(self._d_fake_source())
        fake = fake.detach()
        real = self.real_A if self.opt.no_cgan else networks.cat_pair(self.real_B, self.real_A)
        n = self.n_netD2
        self.loss_D2, each = self._gan([(d, fake, False) for d in self.netD2] + [(d, real, True) for d in self.netD2], [0.5] * (2 * n))
        self.loss_D2_fake, self.loss_D2_real = each[:n].sum(), each[n:].sum()
        self._backward(self.loss_D2)

    def backward_G(self):
        """(:224-259)"""
        o = self.opt
        for netD in self.netD2:
            netD.compute_param_grads = not getattr(o, 'skip_wasted_D_wgrad', False)
        self.loss_G2_GAN, _ = self._gan([(d, self._d_fake_source(), True) for d in self.netD2], list(o.lambda_D2))
        for netD in self.netD2:
            netD.compute_param_grads = True
        self.loss_G1_CE = SegmentationModel.compute_cross_entropy_loss(self, weighted=True)
        self.loss_G_L1 = self.criterionL1(self.fake_B, self.real_B)
        self.loss_G_cycle = self.criterionL1(self.recon_A, self.real_A)
        self.loss_G = self.loss_G1_CE * o.lambda_A + self.loss_G2_GAN + self.loss_G_L1 * o.lambda_B + self.loss_G_cycle * o.lambda_A_cycle
        self._backward(self.loss_G)

    def optimize_parameters(self):
        ops.begin_step()      # one launch zeroes every statistics arena of the step
        o = self.opt
        self.forward()
        for n_up, opt_, back in ((o.n_update_D2, self.optimizer_D2, self.backward_D2), (o.n_update_G, self.optimizer_G, self.backward_G)):
            for _ in range(n_up):
                opt_.zero_grad()
                back()
                if self.grad_sync is not None:
                    self.grad_sync(opt_)
                opt_.step()
                if n_up > 1:
                    self.sample_noise()

    def get_current_errors(self):
        return OrderedDict([('G_CE', float(self.loss_G1_CE.detach())), ('G_GAN', float(self.loss_G2_GAN.detach())),
                            ('G_L1', float(self.loss_G_L1.detach())), ('G_cycle', float(self.loss_G_cycle.detach())),
                            ('D_real', float(self.loss_D2_real)), ('D_fake', float(self.loss_D2_fake))])

    def get_current_visuals(self, save_as_single_image=False):
        three = lambda t: t if t.shape[1] in (1, 3) else torch.cat([t, torch.zeros_like(t[:, :1])], 1)[:, :3]      # noqa: E731
        return OrderedDict([('image', self.real_A.detach()), ('label', three(self.real_B.detach() * 2 - 1)),
                            ('prediction', three(self.fake_B.detach() * 2 - 1))])

    def save(self, label):
        self.save_network(self.netG1, 'G1', label, gpu_ids=self.gpu_ids)
        self.save_network(self.netG2, 'G2', label, gpu_ids=self.gpu_ids)
        for n, netD in enumerate(self.netD2):
            self.save_network(netD, 'D2_%d' % n, label, gpu_ids=self.gpu_ids)

    def update_learning_rate(self):
        """(:313-332): G1 follows lr1, G2 and D2 follow lr2."""
        nd = self.opt.niter_decay
        lr, lr1, lr2 = (max(0, old - base / nd) for old, base in ((self.old_lr, self.opt.lr), (self.old_lr1, self.opt.lr1), (self.old_lr2, self.opt.lr2)))
        for g in self.optimizer_D2.param_groups:
            g['lr'] = lr2
        for g in self.optimizer_G.param_groups:
            g['lr'] = lr1 if g.get('name') == 'G1' else lr2 if g.get('name') == 'G2' else lr
        self.optimizer_D2.sync_lr()
        self.optimizer_G.sync_lr()
        print('update learning rate: %f -> %f, %f -> %f' % (self.old_lr1, lr1, self.old_lr2, lr2))
        self.old_lr, self.old_lr1, self.old_lr2 = lr, lr1, lr2

    reset_accs = SegmentationModel.reset_accs
    accum_accs = SegmentationModel.accum_accs
    compute_current_Rand_score = SegmentationModel.compute_current_Rand_score
    compute_current_accuracy = SegmentationModel.compute_current_accuracy
    get_current_accs = SegmentationModel.get_current_accs
