"""TwoStageCycleModel (models/twostage_cycle_model.py:14-500), the DSGAN trainer: G1 (noise -> label), G2 (label, noise ->
image), F2 (image -> label), discriminators D1 on labels and D2 on (label, image) pairs; the step is D1, then D2, then
one joint G update with  G1_GAN + G2_GAN / num_pairs + lambda_A L1 + lambda_B BCE(F2(real_B)) + lambda_A_cycle BCE(recon_real_A)
+ lambda_A_cycle lambda_fake_cycle BCE(recon_fake_A)  (:405-409), driving the MI355X kernels.

Same method names, loss definitions and update order as the reference.  Implemented: the binary GAN objective and the
`--use_multi_class_GAN` 3-way head (cross-entropy on PyTorch's kernels: 3 x 67 x 67 maps), `--transform_1to2 None | bilinear_2`."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import networks, ops
from .base_model import BaseModel
from .image_pool import ImagePool
from .optim import AdamGroups, FusedAdam


class TwoStageCycleModel(BaseModel):
    cycle = True      # TwoStageModel (below) is the same trainer without F2 and the cycle terms

    def name(self):
        return 'TwoStageCycleModel'

    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        self.isTrain = opt.isTrain
        self.multi_class = bool(getattr(opt, 'use_multi_class_GAN', False)) and self.cycle
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
        if self.isTrain and opt.use_fixed_noise1:
            self.fixed_noise1 = self.Tensor(opt.noise_pool_size, opt.noise_nc1, opt.noiseSize1, opt.noiseSize1)
            if self.device.type == 'cuda':
                ops.normal_fill(self.fixed_noise1, self._rng_seed + 31, self._rng_offset)
            else:
                self.fixed_noise1.normal_(0, 1)

        self.netG1 = networks.define_G(opt.input_nc, 0, opt.ngf1, opt.which_model_netG1, opt.norm, not opt.no_dropout1,
                                       n_layers_G=opt.n_layers_G1, use_residual=False, use_fcn=opt.noiseSize1 != 1,
                                       noise_nc=opt.noise_nc1, add_gaussian_noise=opt.add_gaussian_noise,
                                       gaussian_sigma=opt.gaussian_sigma, upsample_mode=opt.upsample_mode1,
                                       n_layers_CRN_block=opt.n_layers_CRN_block1,
                                       share_label_weights=not opt.no_share_label_block_weights1, gpu_ids=self.gpu_ids)
        self.netG2 = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf2, opt.which_model_netG2, opt.norm,
                                       not opt.no_dropout2, n_layers_G=opt.n_layers_G2, use_residual=opt.use_residual2,
                                       use_fcn=False, noise_nc=opt.noise_nc2, add_gaussian_noise=opt.add_gaussian_noise,
                                       gaussian_sigma=opt.gaussian_sigma, upsample_mode=opt.upsample_mode2,
                                       n_layers_CRN_block=opt.n_layers_CRN_block2,
                                       share_label_weights=not opt.no_share_label_block_weights2, gpu_ids=self.gpu_ids)
        self.netF2 = None
        if self.cycle:
            self.netF2 = networks.define_G(opt.output_nc, opt.input_nc, opt.nff2, opt.which_model_netF2, opt.norm,
                                           not opt.no_dropout2, n_layers_G=opt.n_layers_F2, use_residual=opt.use_residual2,
                                           use_fcn=False, noise_nc=opt.noise_nc2, add_gaussian_noise=opt.add_gaussian_noise,
                                           gaussian_sigma=opt.gaussian_sigma, upsample_mode=opt.upsample_mode2,
                                           n_layers_CRN_block=opt.n_layers_CRN_block2,
                                           share_label_weights=not opt.no_share_label_block_weights2, gpu_ids=self.gpu_ids)
        if 'bilinear' in opt.transform_1to2:
            sc = int(opt.transform_1to2.split('_')[1])
            if sc != 2:
                raise NotImplementedError("--transform_1to2 bilinear_%d: only the x2 transform is on the MI355X path" % sc)
            self.transform = networks.bilinear_upsample2x
            self.transform_inverse = lambda x: F.avg_pool2d(x, sc, sc)      # applied to the real label batch only (input prep)
        else:
            self.transform = lambda x: x
            self.transform_inverse = lambda x: x
        if self.isTrain:
            assert (len(opt.scale_factor1) == len(opt.lambda_D1) == len(opt.n_layers_D1))
            assert (len(opt.scale_factor2) == len(opt.lambda_D2) == len(opt.n_layers_D2))
            self.n_netD1, self.n_netD2 = len(opt.scale_factor1), len(opt.scale_factor2)
            self.netD1, self.netD2 = [], []
            for scale, n_layers in zip(opt.scale_factor1, opt.n_layers_D1):
                d = networks.define_D(opt.input_nc, opt.ndf1, opt.which_model_netD1, n_layers_D=n_layers, norm=opt.norm,
                                      use_sigmoid=opt.no_lsgan1, scale_factor=scale, num_classes=2, gpu_ids=self.gpu_ids)
                d.fuse_sigmoid_into_loss = True
                self.netD1.append(d)
            d2_nc = opt.output_nc if opt.no_cgan else opt.output_nc + opt.input_nc
            for scale, n_layers in zip(opt.scale_factor2, opt.n_layers_D2):
                d = networks.define_D(d2_nc, opt.ndf2, opt.which_model_netD2, n_layers_D=n_layers, norm=opt.norm,
                                      use_sigmoid=opt.no_lsgan2, scale_factor=scale, num_classes=3 if self.multi_class else 2,
                                      gpu_ids=self.gpu_ids)
                d.fuse_sigmoid_into_loss = not self.multi_class
                self.netD2.append(d)
            if self.gpu_ids:
                networks.pack_flat(self.netD1)
                networks.pack_flat(self.netD2)
        nets = [('G1', self.netG1), ('G2', self.netG2)] + ([('F2', self.netF2)] if self.cycle else [])
        self._gnets = nets
        if self.isTrain and opt.sequential_train:
            for label, net in nets:
                if label in opt.which_model_to_load:
                    self.load_network(net, label, opt.which_epoch_sequential, model_dir=opt.pretrained_model_dir)
            for tag, ds in (('D1', self.netD1), ('D2', self.netD2)):
                if tag in opt.which_model_to_load:
                    for n, netD in enumerate(ds):
                        self.load_network(netD, '%s_%d' % (tag, n), opt.which_epoch_sequential, model_dir=opt.pretrained_model_dir)
        if not self.isTrain or opt.continue_train:
            for label, net in nets:
                self.load_network(net, label, opt.which_epoch)
            if self.isTrain:
                for tag, ds in (('D1', self.netD1), ('D2', self.netD2)):
                    for n, netD in enumerate(ds):
                        self.load_network(netD, '%s_%d' % (tag, n), opt.which_epoch)

        if self.isTrain:
            self.fake_pool1 = ImagePool(opt.pool_size)
            self.fake_pool2 = ImagePool(opt.pool_size)
            self.fake_pool2_1, self.fake_pool2_2 = ImagePool(opt.pool_size), ImagePool(opt.pool_size)     # multi-class (:122-124)
            if opt.use_fixed_noise1:
                self.noise_pool1 = ImagePool(opt.noise_pool_size)
                self.noise_pool1.query(self.fixed_noise1)
            self.old_lr, self.old_lr1, self.old_lr2 = opt.lr, opt.lr1, opt.lr2
            self.criterionGAN1 = networks.GANLoss(use_lsgan=not opt.no_lsgan1)
            self.criterionGAN2 = (networks.GANLossMultiClass(use_lsgan=not opt.no_lsgan2, num_classes=3) if self.multi_class
                                  else networks.GANLoss(use_lsgan=not opt.no_lsgan2))
            self.criterionL1 = networks.WeightedL1Loss()
            self.backward_D2 = self.backward_D2_multiclass if self.multi_class else self.backward_D2_binary
            groups = [{'name': 'G1', 'params': self.netG1.parameters(), 'lr': opt.lr1},
                      {'name': 'G2', 'params': self.netG2.parameters(), 'lr': opt.lr2}]
            if self.cycle:
                groups.append({'name': 'F2', 'params': self.netF2.parameters(), 'lr': opt.lr2})
            self.optimizer_G = AdamGroups(groups, lr=opt.lr, betas=(opt.beta1, 0.999))
            self.optimizer_D1 = FusedAdam([p for d in self.netD1 for p in d.model.parameters()], lr=opt.lr1, betas=(opt.beta1, 0.999))
            self.optimizer_D2 = FusedAdam([p for d in self.netD2 for p in d.model.parameters()], lr=opt.lr2, betas=(opt.beta1, 0.999))
            self.grad_sync = None
            self._pool_overrides = None     # graphed step: static buffers the host-side ImagePools fill

    # ---- hipGraph hooks (graph_step.GraphedStep) --------------------------------------------------
    def _pool_sources(self):
        """What the step feeds to ImagePool.query, in the reference's order (backward_D1, then backward_D2_binary)."""
        o = self.opt
        srcs = [self.fake_A]
        if self.multi_class:
            return srcs + [self._pair(self.real_A, self.fake_B_from_real_A), self._pair(self.transform(self.fake_A), self.fake_B_from_fake_A)]
        if 'real_fake' in o.GAN_losses_D2:
            srcs.append(self._pair(self.real_A, self.fake_B_from_real_A))
        if 'fake_fake' in o.GAN_losses_D2:
            srcs.append(self._pair(self.transform(self.fake_A), self.fake_B_from_fake_A))
        return srcs

    def _query(self, idx, pool, src):
        if self._pool_overrides is not None:
            return self._pool_overrides[idx]
        return pool.query(src())

    def graph_spec(self):
        o = self.opt
        ups = (o.n_update_D1, o.n_update_D2, o.n_update_G) if self.cycle else (1, 1, 1)
        assert ups == (1, 1, 1) and not o.use_fixed_noise1, "graphed two-stage step: one update each, device-drawn latents"
        npool2 = ('real_fake' in o.GAN_losses_D2) + ('fake_fake' in o.GAN_losses_D2)
        pools2 = [self.fake_pool2_1, self.fake_pool2_2] if self.multi_class else [self.fake_pool2] * npool2
        prog = [[self.optimizer_D1.zero_grad, self.backward_D1], ("sync", self.optimizer_D1),
                [self.optimizer_D1.step, self.optimizer_D2.zero_grad, self.backward_D2], ("sync", self.optimizer_D2),
                [self.optimizer_D2.step, self.optimizer_G.zero_grad, self.backward_G], ("sync", self.optimizer_G),
                [self.optimizer_G.step]]
        return dict(pools=[self.fake_pool1] + pools2, sources=self._pool_sources,
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

    def _generate(self):
        """The six generator calls of forward() / sample_noise() (:193-226)."""
        o = self.opt
        self.fake_A = self.netG1.forward(self.noise1)
        if self.cycle:
            self.fake_A_from_real_B = self.netF2.forward(self.real_B, self.noise2)
        self.fake_B_from_real_A = self.netG2.forward(self.real_A, self.noise2)
        src = self.fake_A.detach() if o.detach_G1_from_G2_x else self.fake_A
        self.fake_B_from_fake_A = self.netG2.forward(self.transform(src), self.noise2)
        if self.cycle:
            self.recon_real_A = self.netF2.forward(self.fake_B_from_real_A, self.noise2)
            self.recon_fake_A = self.netF2.forward(self.fake_B_from_fake_A, self.noise2)

    def forward(self):
        self.real_A, self.real_B = self.input_A, self.input_B
        if self.isTrain and self.opt.use_fixed_noise1:
            self.noise1 = self.noise_pool1.sample(self.opt.batchSize)
        else:
            self.noise1 = self._draw(1).clone()
        self.noise2 = self._draw(2).clone()
        self._generate()

    def sample_noise(self):
        self.noise1 = self._draw(1).clone()
        self.noise2 = self._draw(2).clone()
        self._generate()

    def test(self):
        with torch.no_grad():
            self.noise1, self.noise2 = self._draw(1).clone(), self._draw(2).clone()
            self.fake_A = self.netG1.forward(self.noise1)
            self.fake_B_from_fake_A = self.netG2.forward(self.transform(self.fake_A), self.noise2)

    def get_image_paths(self):
        return self.image_paths

    # ---- losses ---------------------------------------------------------------------------------
    def _gan(self, crit, jobs, weights):
        preds = networks.multi_forward([(d, x) for d, x, _ in jobs])
        return crit.weighted_sum(preds, [r for _, _, r in jobs], weights)

    def backward_D1(self):
        """(:245-262)"""
        fake = self._query(0, self.fake_pool1, lambda: self.fake_A).detach()
        real = self.transform_inverse(self.real_A)
        n = self.n_netD1
        self.loss_D1, each = self._gan(self.criterionGAN1, [(d, fake, False) for d in self.netD1] + [(d, real, True) for d in self.netD1],
                                       [0.5] * (2 * n))
        self.loss_D1_fake, self.loss_D1_real = each[:n].sum(), each[n:].sum()
        self._backward(self.loss_D1)

    def _pair(self, a, b):
        return b if self.opt.no_cgan else networks.cat_pair(a, b)

    def backward_D2_binary(self):
        """(:264-299) -- one ImagePool serves both fake pairs, queried in the reference's order."""
        o = self.opt
        jobs, n = [], self.n_netD2
        fakes = []
        if 'real_fake' in o.GAN_losses_D2:
            fakes.append(self._query(1, self.fake_pool2, lambda: self._pair(self.real_A, self.fake_B_from_real_A)).detach())
        if 'fake_fake' in o.GAN_losses_D2:
            fakes.append(self._query(1 + len(fakes), self.fake_pool2,
                                     lambda: self._pair(self.transform(self.fake_A), self.fake_B_from_fake_A)).detach())
        num_fake_pairs = len(fakes)
        for f in fakes:
            jobs += [(d, f, False) for d in self.netD2]
        real = self._pair(self.real_A, self.real_B)
        jobs += [(d, real, True) for d in self.netD2]
        weights = [0.5 / num_fake_pairs] * (n * num_fake_pairs) + [0.5] * n
        total, each = None, []
        for i0 in range(0, len(jobs), 8):           # the fused loss node takes <= 8 terms
            t, e = self._gan(self.criterionGAN2, jobs[i0:i0 + 8], weights[i0:i0 + 8])
            total = t if total is None else total + t
            each.append(e)
        each = torch.cat(each)
        self.loss_D2_fake = each[:n * num_fake_pairs].sum() / num_fake_pairs
        self.loss_D2_real = each[n * num_fake_pairs:].sum()
        self.loss_D2 = total
        self._backward(self.loss_D2)

    def backward_D2_multiclass(self):
        """(:302-335): classes 0 = (real_A, real_B), 1 = (real_A, fake_B), 2 = (fake_A, fake_B); cross-entropy; one pool per fake class."""
        real = self._pair(self.real_A, self.real_B)
        f1 = self._query(1, self.fake_pool2_1, lambda: self._pair(self.real_A, self.fake_B_from_real_A)).detach()
        f2 = self._query(2, self.fake_pool2_2, lambda: self._pair(self.transform(self.fake_A), self.fake_B_from_fake_A)).detach()
        n = self.n_netD2
        preds = networks.multi_forward([(d, real) for d in self.netD2] + [(d, f1) for d in self.netD2] + [(d, f2) for d in self.netD2])
        ce = [sum(self.criterionGAN2(p, k) for p in preds[k * n:(k + 1) * n]) for k in range(3)]
        self.loss_D2_0, self.loss_D2_1, self.loss_D2_2 = ce
        self.loss_D2 = (ce[0] + ce[1] + ce[2]) / 3
        self.loss_D2_real, self.loss_D2_fake = ce[0], (ce[1] + ce[2]) / 2
        self._backward(self.loss_D2)

    def _g2_gan_term(self, fake, label, trick):
        """sum_i lambda_D2[i] * GAN(D2_i(fake), real) of one (label, image) pair (:348-367); `label` is a thunk the factD variant calls."""
        o = self.opt
        t, _ = self._gan(self.criterionGAN2, [(d, fake, trick) for d in self.netD2], [l if trick else -l for l in o.lambda_D2])
        return t

    def backward_G(self):
        """(:337-410)"""
        o = self.opt
        for netD in self.netD1 + self.netD2:
            netD.compute_param_grads = not getattr(o, 'skip_wasted_D_wgrad', False)
        trick = not o.no_logD_trick
        self.loss_G1_GAN, _ = self._gan(self.criterionGAN1, [(d, self.fake_A, trick) for d in self.netD1],
                                        [l if trick else -l for l in o.lambda_D1])
        pairs, labels = [], []          # `labels`: the half-size label of each pair (what the factD variant shows its D1)
        if 'real_fake' in o.GAN_losses_G2:
            pairs.append(self._pair(self.real_A, self.fake_B_from_real_A))
            labels.append(lambda: self.transform_inverse(self.real_A))
        if 'fake_fake' in o.GAN_losses_G2:
            fa = self.fake_A.detach() if o.detach_G1_from_G2_y else self.fake_A
            pairs.append(self._pair(self.transform(fa), self.fake_B_from_fake_A))
            labels.append(lambda: fa)
        num_fake_pairs = len(pairs)
        self.loss_G2_GAN = 0
        for fake, label in zip(pairs, labels):
            if self.multi_class:       # flipped_label = 0 (:352); criterionGAN2(pred, False) addresses class 0 as well
                preds = networks.multi_forward([(d, fake) for d in self.netD2])
                for p, lam in zip(preds, o.lambda_D2):
                    self.loss_G2_GAN = self.loss_G2_GAN + self.criterionGAN2(p, 0) * (lam if trick else -lam)
                continue
            self.loss_G2_GAN = self.loss_G2_GAN + self._g2_gan_term(fake, label, trick)
        for netD in self.netD1 + self.netD2:
            netD.compute_param_grads = True
        if 'real_fake' in o.GAN_losses_G2:
            self.loss_G2_L1 = self.criterionL1.from_labels(self.fake_B_from_real_A, self.real_B, self.real_A,
                                                           o.weights if self.cycle else None, 1.0)
        else:
            self.loss_G2_L1 = 0
        if not self.cycle:      # TwoStageModel.backward_G (twostage_model.py:369-377): plain L1Loss, lambda_G1 / lambda_G2
            self.loss_G = self.loss_G1_GAN * o.lambda_G1 + self.loss_G2_GAN / num_fake_pairs * o.lambda_G2 \
                + self.loss_G2_L1 * o.lambda_G2 * o.lambda_A
            self._backward(self.loss_G)
            return
        self.loss_F2_CE = networks.bce_on_rescaled(self.fake_A_from_real_B, self.real_A)
        self.loss_G2_real_cycle = networks.bce_on_rescaled(self.recon_real_A, self.real_A)
        self.loss_G2_fake_cycle = networks.bce_on_rescaled(self.recon_fake_A, self.transform(self.fake_A.detach()))
        self.loss_G = self.loss_G1_GAN + self.loss_G2_GAN / num_fake_pairs + self.loss_G2_L1 * o.lambda_A \
            + self.loss_F2_CE * o.lambda_B + self.loss_G2_real_cycle * o.lambda_A_cycle \
            + self.loss_G2_fake_cycle * o.lambda_A_cycle * o.lambda_fake_cycle
        self._backward(self.loss_G)

    def optimize_parameters(self):
        ops.begin_step()      # one launch zeroes every statistics arena of the step
        o = self.opt
        self.forward()
        ups = (o.n_update_D1, o.n_update_D2, o.n_update_G) if self.cycle else (1, 1, 1)     # twostage_model.py:379-395: one each
        for n_up, opt_, back in ((ups[0], self.optimizer_D1, self.backward_D1),
                                 (ups[1], self.optimizer_D2, self.backward_D2),
                                 (ups[2], self.optimizer_G, self.backward_G)):
            for _ in range(n_up):
                opt_.zero_grad()
                back()
                if self.grad_sync is not None:
                    self.grad_sync(opt_)
                opt_.step()
                if n_up > 1:
                    self.sample_noise()

    def get_current_errors(self):
        f = lambda v: float(v.detach()) if torch.is_tensor(v) else float(v)
        if not self.cycle:
            return OrderedDict([('G2_GAN', f(self.loss_G2_GAN)), ('D2', f(self.loss_D2)), ('G1_GAN', f(self.loss_G1_GAN)),
                                ('D1', f(self.loss_D1))])
        return OrderedDict([('G2_GAN', f(self.loss_G2_GAN)), ('G2_real_cycle', f(self.loss_G2_real_cycle)),
                            ('G2_fake_cycle', f(self.loss_G2_fake_cycle)), ('D2', f(self.loss_D2)),
                            ('G1_GAN', f(self.loss_G1_GAN)), ('D1', f(self.loss_D1))])

    def get_current_visuals(self, save_as_single_image=False):
        out = OrderedDict([('fake_A', self.transform(self.fake_A).detach()), ('fake_B_fake_A', self.fake_B_from_fake_A.detach())])
        if self.isTrain:
            out.update([('real_A', self.real_A), ('fake_B_real_A', self.fake_B_from_real_A.detach()), ('real_B', self.real_B)])
            if self.cycle:
                out.update([('fake_A_real_B', self.fake_A_from_real_B.detach()), ('recon_real_A', self.recon_real_A.detach()),
                            ('recon_fake_A', self.recon_fake_A.detach())])
        return out

    def save(self, label):
        for tag, net in self._gnets:
            self.save_network(net, tag, label, gpu_ids=self.gpu_ids)
        for tag, ds in (('D1', self.netD1), ('D2', self.netD2)):
            for n, netD in enumerate(ds):
                self.save_network(netD, '%s_%d' % (tag, n), label, gpu_ids=self.gpu_ids)

    def update_learning_rate(self):
        """(:477-500)"""
        o = self.opt
        lr = max(0, self.old_lr - o.lr / o.niter_decay)
        lr1 = max(0, self.old_lr1 - o.lr1 / o.niter_decay)
        lr2 = max(0, self.old_lr2 - o.lr2 / o.niter_decay)
        for g in self.optimizer_D1.param_groups:
            g['lr'] = lr1
        for g in self.optimizer_D2.param_groups:
            g['lr'] = lr2
        for g in self.optimizer_G.param_groups:
            g['lr'] = {'G1': lr1, 'G2': lr2, 'F2': lr2}.get(g.get('name'), lr)
        for opt_ in (self.optimizer_D1, self.optimizer_D2, self.optimizer_G):
            opt_.sync_lr()
        print('update learning rate: %f -> %f, %f -> %f' % (self.old_lr1, lr1, self.old_lr2, lr2))
        self.old_lr, self.old_lr1, self.old_lr2 = lr, lr1, lr2


class TwoStageModel(TwoStageCycleModel):
    """TwoStageModel (models/twostage_model.py:14-448): G1 + G2 with D1 / D2, no label reconstructor and no cycle terms;
    loss_G = lambda_G1 G1_GAN + lambda_G2 (G2_GAN / num_pairs + lambda_A L1)."""
    cycle = False

    def name(self):
        return 'TwoStageModel'


class TwoStageFactDModel(TwoStageModel):
    """TwoStageModel of models/twostage_factD_model.py (`--model twostage_factd`): the image discriminators are factored -- every D2_i
    prediction is multiplied by the prediction of D1_i on the half-size label, upsampled x2 (`transform`) and reflection-padded to
    D2_i's map size (util.mul, util/util.py:131-145) -- in the D2 step (:256-296) and in the generators' GAN term (:352-383).
    The products are formed on probabilities (or raw lsgan scores), so these terms run un-fused: one discriminator call each, the
    upsample / pad / product / loss on PyTorch's kernels over the 35 x 35-sized maps."""

    def name(self):
        return 'TwoStageFactDModel'

    def initialize(self, opt):
        assert not getattr(opt, 'use_multi_class_GAN', False) and not getattr(opt, 'no_cgan', False)          # twostage_factD_model.py:23-24
        TwoStageModel.initialize(self, opt)
        if self.isTrain:
            assert self.n_netD1 == self.n_netD2, "factored discriminators come in (D1_i, D2_i) pairs"

    @staticmethod
    def _mul(in1, in2):
        """util.mul (util/util.py:131-145): in1 reflection-padded up to in2's size; the reference returns None when in1 is larger."""
        if in1.shape == in2.shape:
            return in1 * in2
        if not (in1.shape[2] <= in2.shape[2] and in1.shape[3] <= in2.shape[3]):
            raise ValueError("twostage_factd: the upsampled D1 map %s is larger than D2's %s (the reference's util.mul returns None here); "
                             "choose --n_layers_D1 / --n_layers_D2 so that it is not" % (tuple(in1.shape[2:]), tuple(in2.shape[2:])))
        pl, pb = int((in2.shape[3] - in1.shape[3]) / 2), int((in2.shape[2] - in1.shape[2]) / 2)
        pr, pt = in2.shape[3] - in1.shape[3] - pl, in2.shape[2] - in1.shape[2] - pb
        return F.pad(in1, (pl, pr, pt, pb), mode='reflect') * in2

    def _plain(self, netD, x):
        """One discriminator call returning probabilities (--no_lsgan) or raw scores, never the logits-tagged fused form."""
        fused, netD.fuse_sigmoid_into_loss = netD.fuse_sigmoid_into_loss, False
        try:
            return netD.forward(x)
        finally:
            netD.fuse_sigmoid_into_loss = fused

    def _factored(self, i, label, pair):
        p1 = F.interpolate(self._plain(self.netD1[i], label), scale_factor=2, mode='bilinear', align_corners=False)      # self.transform
        return self._mul(p1, self._plain(self.netD2[i], pair))

    def _crit(self, pred, real):
        if self.opt.no_lsgan2:
            return F.binary_cross_entropy(pred, torch.full_like(pred, 1.0 if real else 0.0))
        return F.mse_loss(pred, torch.full_like(pred, 1.0 if real else 0.0))

    def backward_D2_binary(self):
        """(twostage_factD_model.py:256-296): the label half of every (pooled) pair goes through transform_inverse to its D1."""
        o = self.opt
        n, nc = self.n_netD2, o.input_nc
        fakes = []
        if 'real_fake' in o.GAN_losses_D2:
            fakes.append(self._query(1, self.fake_pool2, lambda: self._pair(self.real_A, self.fake_B_from_real_A)).detach())
        if 'fake_fake' in o.GAN_losses_D2:
            fakes.append(self._query(1 + len(fakes), self.fake_pool2,
                                     lambda: self._pair(self.transform(self.fake_A), self.fake_B_from_fake_A)).detach())
        self.loss_D2_fake = 0
        for f in fakes:
            lab = self.transform_inverse(f.narrow(1, 0, nc)).detach()
            self.loss_D2_fake = self.loss_D2_fake + sum(self._crit(self._factored(i, lab, f), False) for i in range(n))
        self.loss_D2_fake = self.loss_D2_fake / len(fakes)
        real = self._pair(self.real_A, self.real_B)
        lab = self.transform_inverse(self.real_A)
        self.loss_D2_real = sum(self._crit(self._factored(i, lab, real), True) for i in range(n))
        self.loss_D2 = (self.loss_D2_fake + self.loss_D2_real) * 0.5
        self._backward(self.loss_D2)

    def _g2_gan_term(self, fake, label, trick):
        lab = label()
        t = 0
        for i, lam in enumerate(self.opt.lambda_D2):
            pred = self._factored(i, lab, fake)
            t = t + (self._crit(pred, True) * lam if trick else -self._crit(pred, False) * lam)
        return t
