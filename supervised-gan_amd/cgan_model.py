"""CGANModel (models/cgan_model.py:14-258): the conditional GAN trainer -- real_A -> U-Net G -> fake_B, multi-scale
PatchGAN discriminators on cat(real_A, B), GAN + weighted L1 loss, Adam, checkpoints, linear LR decay -- driving the
MI355X kernels.  Same method names, loss definitions and update order as the reference.

Deviations, all observable-behaviour preserving:
  * the per-discriminator loss terms of one step are one fused loss node (GANLoss.weighted_sum); same-architecture
    discriminator calls (the fake and the real batch of one D) run as grouped kernels;
  * the L1 weight map (cgan_model.py:198-207) is evaluated inside the L1 kernel;
  * dropout masks / Gaussian noise in G come from a counter-based Philox kernel, not torch's RNG stream;
  * `--skip_wasted_D_wgrad` as in FCGANModel."""
from collections import OrderedDict

import torch

from . import networks, ops
from .base_model import BaseModel
from .image_pool import ImagePool
from .optim import FusedAdam


class CGANModel(BaseModel):
    def name(self):
        return 'cGANModel'

    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        self.isTrain = opt.isTrain
        # which_channel 'rg_b' means rg --> b (cgan_model.py:33-43)
        idx_dict = {'r': 0, 'g': 1, 'b': 2}
        self.chnl_idx_input = [[idx_dict[c] for c in s] for s in opt.which_channel.split('_')]
        assert len(self.chnl_idx_input) == 2
        opt.input_nc = len(self.chnl_idx_input[0])
        opt.output_nc = self._output_channels(opt)
        self._chnl_dev = [torch.tensor(ix, dtype=torch.long, device=self.device) for ix in self.chnl_idx_input]
        if 'bilinear' in opt.transform_1to2:
            raise NotImplementedError("--transform_1to2 bilinear_* is a test-time option outside the MI355X training path")

        self.input_A = self.Tensor(opt.batchSize, opt.input_nc, opt.fineSize, opt.fineSize)
        self.input_B = self.Tensor(opt.batchSize, opt.output_nc, opt.fineSize, opt.fineSize)
        self.noise = None
        self.noise_ = self.Tensor(opt.batchSize, opt.noise_nc, opt.noiseSize, opt.noiseSize)
        self._rng_seed = 0 if opt.manualSeed is None else int(opt.manualSeed)
        self._rng_offset = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.noise_source = None    # optional callable() -> z tensor (tests inject latents)

        self.netG = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, opt.which_model_netG, opt.norm,
                                      not opt.no_dropout, n_layers_G=opt.n_layers_G, use_residual=opt.use_residual,
                                      use_fcn=opt.noiseSize != 1, noise_nc=opt.noise_nc,
                                      add_gaussian_noise=opt.add_gaussian_noise, gaussian_sigma=opt.gaussian_sigma,
                                      upsample_mode=opt.upsample_mode, n_layers_CRN_block=opt.n_layers_CRN_block,
                                      share_label_weights=not opt.no_share_label_block_weights,
                                      n_layers_G_skip=opt.n_layers_G_skip, gpu_ids=self.gpu_ids)
        if hasattr(self.netG, '_rng_seed'):
            self.netG._rng_seed = 0 if opt.manualSeed is None else int(opt.manualSeed)
        if self.isTrain:
            use_sigmoid = opt.no_lsgan
            assert (len(opt.scale_factor) == len(opt.lambda_D) == len(opt.n_layers_D))
            self.n_netD = len(opt.scale_factor)
            self.netD = []
            netD_input_nc = opt.output_nc if opt.no_cgan else opt.output_nc + opt.input_nc
            for scale, n_layers in zip(opt.scale_factor, opt.n_layers_D):
                d = networks.define_D(netD_input_nc, opt.ndf, opt.which_model_netD, n_layers_D=n_layers, norm=opt.norm,
                                      use_sigmoid=use_sigmoid, scale_factor=scale, gpu_ids=self.gpu_ids)
                d.fuse_sigmoid_into_loss = True
                self.netD.append(d)
            if self.gpu_ids:
                networks.pack_flat(self.netD)
        if not self.isTrain or opt.continue_train:
            self.load_network(self.netG, 'G', opt.which_epoch)
            if self.isTrain:
                for netD, n in zip(self.netD, range(self.n_netD)):
                    self.load_network(netD, 'D_%d' % n, opt.which_epoch)

        if self.isTrain:
            self.fake_pool = ImagePool(opt.pool_size)
            self.old_lr = opt.lr
            self.criterionGAN = networks.GANLoss(use_lsgan=not opt.no_lsgan)
            self.criterionL1 = networks.WeightedL1Loss()
            self.optimizer_G = FusedAdam(self.netG.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999), zero_grads_in_step=True)
            params = []
            for netD in self.netD:
                params += list(netD.model.parameters())
            self.optimizer_D = FusedAdam(params, lr=opt.lr, betas=(opt.beta1, 0.999))
            self.grad_sync = None
            self._pool_override = None

    def _output_channels(self, opt):
        """Channels the generator emits / the discriminators see beside real_A (the segmentation trainer emits class scores)."""
        return len(self.chnl_idx_input[1])

    # ---- data ---------------------------------------------------------------------------------
    def set_input(self, input):
        AtoB = self.opt.which_direction == 'AtoB'
        if self.opt.dataset_mode == 'aligned':
            a, b = input['A' if AtoB else 'B'], input['B' if AtoB else 'A']
        elif self.opt.dataset_mode == 'single':
            a = b = input['A']
        else:
            raise NotImplementedError('Dataset mode [%s] is not recognized' % self.opt.dataset_mode)
        if not (self._gather_input(a, 0, 'input_A') and self._gather_input(b, 1, 'input_B')):
            a = a.to(self.device, non_blocking=True).index_select(1, self._chnl_dev[0])
            b = b.to(self.device, non_blocking=True).index_select(1, self._chnl_dev[1])
            self.input_A.resize_(a.size()).copy_(a)
            self.input_B.resize_(b.size()).copy_(b)
        self.image_paths = input.get('A_paths' if AtoB else 'B_paths')

    def _gather_input(self, data, which, name):
        """One gather kernel per image: reads the batch where it lies (pinned host memory: the kernel is the H2D copy), picks the
        contiguous channel range of --which_channel and writes the padded NHWC buffer behind `self.<name>`.  False: not a case
        the kernel covers (the caller takes the index_select path)."""
        idx = self.chnl_idx_input[which]
        run = idx == list(range(idx[0], idx[0] + len(idx)))
        if not (self.device.type == 'cuda' and run and len(idx) <= 4 and data.dim() == 4 and data.shape[0] == 1
                and data.dtype == torch.float32 and data.stride(3) == 1 and (data.is_cuda or data.is_pinned())):
            return False
        _, _, H, W = data.shape
        buf = getattr(self, '_buf_' + name, None)
        if buf is None or buf.shape[:2] != (H, W):
            buf = torch.zeros((H, W, 4), dtype=torch.float32, device=self.device)
            setattr(self, '_buf_' + name, buf)
            setattr(self, name, ops.logical_view(buf, len(idx)))
        ops.host_batch_to_nhwc(data[:, idx[0]: idx[0] + len(idx)], buf)
        return True

    def _draw_noise(self):
        """z ~ N(0, 1) [1, noise_nc, noiseSize, noiseSize] (cgan_model.py:137-138); only the CRN generator reads it."""
        if not hasattr(self.netG, 'noise_nc'):
            return None
        if self.noise_source is not None:
            self.noise_.copy_(self.noise_source())
        else:
            ops.normal_fill(self.noise_, self._rng_seed + 977, self._rng_offset)
        return self.noise_

    def forward(self):
        self.real_A = self.input_A
        self.real_B = self.input_B
        self.noise = self._draw_noise()
        self.fake_B = self.netG.forward(self.real_A, self.noise)

    sample_noise = forward

    def _pool_source(self):
        """What the reference hands to ImagePool.query (cgan_model.py:160-163)."""
        return self.fake_B if self.opt.no_cgan else networks.cat_pair(self.real_A, self.fake_B)

    def test(self):
        with torch.no_grad():
            self.real_A = self.input_A
            self.noise = self._draw_noise()
            self.fake_B = self.netG.forward(self.real_A, self.noise)

    def get_image_paths(self):
        return self.image_paths

    # ---- losses ---------------------------------------------------------------------------------
    def _d_losses(self, jobs, weights):
        preds = networks.multi_forward([(d, x) for d, x, _ in jobs])
        return self.criterionGAN.weighted_sum(preds, [r for _, _, r in jobs], weights)

    def backward_D(self):
        """loss_D = 0.5 * (sum_i GAN(D_i(fake), 0) + sum_i GAN(D_i(real), 1))   (cgan_model.py:158-182)"""
        fake = self._pool_override if self._pool_override is not None else self.fake_pool.query(self._pool_source())
        fake = fake.detach()
        real = self.real_B if self.opt.no_cgan else networks.cat_pair(self.real_A, self.real_B)
        n = self.n_netD
        self.loss_D, self._each_D = self._d_losses([(d, fake, False) for d in self.netD] + [(d, real, True) for d in self.netD],
                                                   [0.5] * (2 * n))
        self._backward(self.loss_D)

    def backward_G(self):
        """loss_G = sum_i lambda_i * GAN(D_i(cat(A, fake_B)), 1) + lambda_A * L1_w(fake_B, real_B)  (cgan_model.py:184-210)"""
        skip = getattr(self.opt, 'skip_wasted_D_wgrad', False)
        for netD in self.netD:
            netD.compute_param_grads = not skip
        fake = self.fake_B if self.opt.no_cgan else networks.cat_pair(self.real_A, self.fake_B)
        trick = not self.opt.no_logD_trick
        gan, self._each_G = self._d_losses([(d, fake, trick) for d in self.netD],
                                           [l if trick else -l for l in self.opt.lambda_D])
        for netD in self.netD:
            netD.compute_param_grads = True
        self.loss_G_L1 = self.criterionL1.from_labels(self.fake_B, self.real_B, self.real_A, self.opt.weights, self.opt.lambda_A)
        self.loss_G = gan + self.loss_G_L1
        self._backward(self.loss_G)

    @property
    def loss_D_fake(self):
        return self._each_D[:self.n_netD].sum()

    @property
    def loss_D_real(self):
        return self._each_D[self.n_netD:].sum()

    def optimize_parameters(self):
        ops.begin_step(self.optimizer_D.take_zeroing())      # one launch zeroes every statistics arena of the step and D's gradients
        self.forward()
        for _ in range(self.opt.n_update_D):
            self.optimizer_D.zero_grad()
            self.backward_D()
            if self.grad_sync is not None:
                self.grad_sync(self.optimizer_D)
            self.optimizer_D.step()
            if self.opt.n_update_D > 1:
                self.sample_noise()
        for _ in range(self.opt.n_update_G):
            self.optimizer_G.zero_grad()
            self.backward_G()
            if self.grad_sync is not None:
                self.grad_sync(self.optimizer_G)
            self.optimizer_G.step()
            if self.opt.n_update_G > 1:
                self.sample_noise()

    def get_current_errors(self):
        return OrderedDict([('G_GAN', float(self.loss_G.detach())), ('G_L1', float(self.loss_G_L1.detach())),
                            ('D_real', float(self.loss_D_real)), ('D_fake', float(self.loss_D_fake))])

    def get_current_visuals(self, save_as_single_image=False):
        out = OrderedDict([('real_A', self.real_A.detach()), ('fake_B', self.fake_B.detach())])
        if self.isTrain:
            out['real_B'] = self.real_B.detach()
        return out

    def save(self, label):
        self.save_network(self.netG, 'G', label, gpu_ids=self.gpu_ids)
        for netD, n in zip(self.netD, range(self.n_netD)):
            self.save_network(netD, 'D_%d' % n, label, self.gpu_ids)

    def update_learning_rate(self):
        lrd = self.opt.lr / self.opt.niter_decay
        lr = self.old_lr - lrd
        for opt_ in (self.optimizer_D, self.optimizer_G):
            for param_group in opt_.param_groups:
                param_group['lr'] = lr
            opt_.sync_lr()
        print('update learning rate: %f -> %f' % (self.old_lr, lr))
        self.old_lr = lr
