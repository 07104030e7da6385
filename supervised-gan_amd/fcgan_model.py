"""FCGANModel (models/fcgan_model.py:26-236): the unconditional GAN trainer -- noise -> G -> fake,
multi-scale discriminator list, D step / G step, Adam, checkpoint save/load, linear LR decay --
driving the MI355X kernels.  Method names, loss definitions and update order follow the reference
line by line in *behaviour*; data stays device resident and every network call is one autograd node.

Deviations, all observable-behaviour preserving:
  * sigmoid+BCE is evaluated by one kernel on the discriminator logits (GANLoss);
  * `--skip_wasted_D_wgrad`: the reference computes discriminator weight gradients during the G step
    and zeroes them before they are ever used (fcgan_model.py:176,182); with the flag they are not
    computed;
  * latent noise comes from a counter-based Philox kernel (`normal_fill`), not torch's RNG stream."""
from collections import OrderedDict

import torch

from . import networks, ops
from .base_model import BaseModel
from .image_pool import ImagePool
from .optim import FusedAdam


class FCGANModel(BaseModel):
    def name(self):
        return 'FCGANModel'

    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        self.isTrain = opt.isTrain
        # parse which_channel (fcgan_model.py:47-58)
        idx_dict = {'r': 0, 'g': 1, 'b': 2}
        self.chnl_idx_input, self.chnl_idx_visual = [], []
        for s in opt.which_channel.split('_'):
            self.chnl_idx_visual.append([idx_dict[c] for c in s])
            self.chnl_idx_input += [idx_dict[c] for c in s]
        opt.input_nc = len(self.chnl_idx_input)
        self._chnl_dev = torch.tensor(self.chnl_idx_input, dtype=torch.long, device=self.device)

        zshape = (opt.batchSize, opt.noise_nc, opt.noiseSize, opt.noiseSize)
        self.input = self.Tensor(opt.batchSize, opt.input_nc, opt.fineSize, opt.fineSize)
        self.noise = None
        if self.device.type == 'cuda' and opt.batchSize == 1:
            # the latent lives in the padded NHWC buffer the generator reads; `noise_` is its logical [1, C, H, W] view
            self._noise_buf = torch.zeros((opt.noiseSize, opt.noiseSize, ops.pad4(opt.noise_nc)), dtype=torch.float32, device=self.device)
            self.noise_ = ops.logical_view(self._noise_buf, opt.noise_nc)
        else:
            self._noise_buf = None
            self.noise_ = self.Tensor(*zshape)
        self._rng_seed = 0 if opt.manualSeed is None else int(opt.manualSeed)
        self._rng_offset = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.noise_source = None    # optional callable() -> z tensor (tests inject latents)
        if self.device.type == 'cuda':
            self.fixed_noiseA = self._draw_noise().clone()
            self.fixed_noiseB = self._draw_noise().clone()
        else:   # a CPU-resident model can be built (checkpoint surgery, tests) but never runs a network
            self.fixed_noiseA = torch.randn(zshape)
            self.fixed_noiseB = torch.randn(zshape)

        self.netG = networks.define_G(opt.input_nc, 0, opt.ngf, opt.which_model_netG, opt.norm, not opt.no_dropout,
                                      n_layers_G=opt.n_layers_G, use_residual=opt.use_residual,
                                      use_fcn=opt.noiseSize != 1, noise_nc=opt.noise_nc,
                                      add_gaussian_noise=opt.add_gaussian_noise, gaussian_sigma=opt.gaussian_sigma,
                                      upsample_mode=opt.upsample_mode, n_layers_CRN_block=opt.n_layers_CRN_block,
                                      share_label_weights=not opt.no_share_label_block_weights, gpu_ids=self.gpu_ids)
        if self.isTrain:
            use_sigmoid = opt.no_lsgan
            assert (len(opt.scale_factor) == len(opt.lambda_D) == len(opt.n_layers_D))
            self.n_netD = len(opt.scale_factor)
            self.netD = []
            for scale, n_layers in zip(opt.scale_factor, opt.n_layers_D):
                d = networks.define_D(opt.input_nc, opt.ndf, opt.which_model_netD, n_layers_D=n_layers, norm=opt.norm,
                                      use_sigmoid=use_sigmoid, scale_factor=scale, gpu_ids=self.gpu_ids)
                d.fuse_sigmoid_into_loss = True
                self.netD.append(d)
            if self.gpu_ids:
                networks.pack_flat(self.netD)   # one arena: single Adam segment, single gradient all-reduce
        if not self.isTrain or opt.continue_train:
            self.load_network(self.netG, 'G', opt.which_epoch)
            if self.isTrain:
                for netD, n in zip(self.netD, range(self.n_netD)):
                    self.load_network(netD, 'D_%d' % n, opt.which_epoch)

        if self.isTrain:
            self.fake_pool = ImagePool(opt.pool_size)
            self.old_lr = opt.lr
            self.criterionGAN = networks.GANLoss(use_lsgan=not opt.no_lsgan)
            self.optimizer_G = FusedAdam(self.netG.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999), zero_grads_in_step=True)
            params = []
            for netD in self.netD:
                params += list(netD.model.parameters())   # "all learnable parameters should be in netD.model"
            self.optimizer_D = FusedAdam(params, lr=opt.lr, betas=(opt.beta1, 0.999))
            self.grad_sync = None   # data-parallel hook: callable(optimizer) run between backward and step
            self._pool_override = None   # graphed step: static buffer the host-side ImagePool fills
            self._group = bool(self.gpu_ids) and not getattr(opt, 'no_group', False)
            n_streams = 2 * self.n_netD if (self.gpu_ids and not getattr(opt, 'no_d_streams', False)) else 0
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(n_streams)]

    # ---- data ---------------------------------------------------------------------------------
    def _draw_noise(self, alt=False):
        """alt: into the second latent buffer (sample_noise_and_prefetch: two latents alive at once)."""
        if alt and getattr(self, '_noise_buf_alt', None) is None:
            self._noise_buf_alt = torch.zeros_like(self._noise_buf)
            self._noise_alt = ops.logical_view(self._noise_buf_alt, self.opt.noise_nc)
        buf, view = (self._noise_buf_alt, self._noise_alt) if alt else (self._noise_buf, self.noise_)
        if self.noise_source is not None:
            z = self.noise_source()
            view.copy_(z)
        elif buf is not None:
            ops.normal_fill_nhwc(buf, self.opt.noise_nc, self._rng_seed, self._rng_offset)
        else:
            ops.normal_fill(view, self._rng_seed, self._rng_offset)
        return view

    def set_input(self, input):
        AorB = self.opt.which_direction == 'A'
        data = input['A' if AorB else 'B']
        idx = self.chnl_idx_input
        run = idx == list(range(idx[0], idx[0] + len(idx)))       # a contiguous channel range ('rg', 'gb', 'r', ...)
        if (self.device.type == 'cuda' and run and data.dim() == 4 and data.shape[0] == 1 and data.dtype == torch.float32
                and data.stride(3) == 1 and (data.is_cuda or data.is_pinned())):
            # one gather kernel reads the batch where it lies (pinned host memory is device-mapped: the kernel IS the H2D copy,
            # queued with the step's kernels), picks the channels and writes the padded NHWC buffer the discriminators read
            _, _, H, W = data.shape
            if getattr(self, '_input_buf', None) is None or self._input_buf.shape[:2] != (H, W):
                self._input_buf = torch.zeros((H, W, 4), dtype=torch.float32, device=self.device)
                self.input = ops.logical_view(self._input_buf, len(idx))
            ops.host_batch_to_nhwc(data[:, idx[0]: idx[0] + len(idx)], self._input_buf)
        else:
            data = data.to(self.device, non_blocking=True).index_select(1, self._chnl_dev)
            self._input_buf = None
            self.input = self.input if self.input.is_contiguous() else torch.empty(0, device=self.device)
            self.input.resize_(data.size()).copy_(data)
        self.image_paths = input.get('A_paths' if AorB else 'B_paths')

    def forward(self):
        self.real = self.input
        self.noise = self._draw_noise()
        self.netG._keep_next = getattr(self, '_prefetch', False)      # graphed step: forward_pair() refills THIS call's buffers
        self.fake = self.netG.forward(self.noise)

    def sample_noise(self):
        self.real = self.input
        self.noise = self._draw_noise()
        self.fake = self.netG.forward(self.noise)

    # ---- the step's last re-draw and the next step's forward() as one pass over the generator (graph_step.GraphedStep) -----------
    def prefetch_supported(self):
        """The re-draw after the last G update (fcgan_model.py:192-193) and the forward() that opens the next step (:179) see the same
        generator weights and independent latents: GraphedStep runs them as ONE two-problem pass (chain.forward_pair) when the
        generator is a plain chain without dropout state and the latents are drawn on the device."""
        from .chain import ChainNet
        o = self.opt
        return (self.isTrain and o.n_update_G > 1 and o.n_update_D == 1 and self.noise_source is None and self._noise_buf is not None
                and isinstance(self.netG, ChainNet) and hasattr(self.netG, '_wrap_output') and type(self.netG).__name__ == 'FCGANGenerator'
                and not any(L.drop > 0 for L in self.netG.layers))

    def sample_noise_and_prefetch(self):
        """sample_noise() of this step, then forward() of the next one: the same two latents in the same order, one launch per layer.
        `fake` is the re-drawn sample (what the reference leaves behind after a step); adopt_prefetched() installs the other one."""
        from .chain import forward_pair
        if getattr(self, '_noise_buf_alt', None) is None:
            self._noise_buf_alt = torch.zeros_like(self._noise_buf)
            self._noise_alt = ops.logical_view(self._noise_buf_alt, self.opt.noise_nc)
        za, zb = self._noise_alt, self.noise_
        full = getattr(self.netG, '_kept_full', None)
        fused = full is not None and full.numel() * 8 <= (1 << 22) and za.numel() <= 65536
        if fused:      # both latents and the cleared statistics arena of the pass in one launch
            ops.normal_fill_nhwc_pair(self._noise_buf_alt, self._noise_buf, self.opt.noise_nc, self._rng_seed, self._rng_offset, full)
        else:
            self._draw_noise(alt=True)
            self._draw_noise()
        self.noise = za
        self.fake, self._fake_next = forward_pair(self.netG, za, zb, arena_zeroed=fused)

    def adopt_prefetched(self):
        """What forward() would have done: `noise`, `fake` of the step that starts now (already computed by sample_noise_and_prefetch)."""
        self.real = self.input
        self.noise = self.noise_
        self.fake = self._fake_next

    def _pool_source(self):
        return self.fake

    def test(self):
        with torch.no_grad():
            self.noise = self._draw_noise()
            self.fake = self.netG.forward(self.noise)

    def get_image_paths(self):
        return self.image_paths

    # ---- losses ---------------------------------------------------------------------------------
    def _d_losses(self, jobs, weights):
        """[(netD, input, target_is_real)], weights -> (total, each): total = sum_i w_i * GANLoss(D_i(x_i), t_i).

        The discriminator chains are independent and individually too small for 256 CUs (a 17x17 layer is 24
        workgroups).  Same-architecture chains run as GROUPED kernels -- one launch per layer for all of them --
        and every loss term plus the scalar arithmetic around it is one forward and one backward kernel.
        Fallback (--no_group or heterogeneous discriminators): one HIP stream per chain, forked from and
        joined back into the current stream."""
        if self._group and networks.can_group([d for d, _, _ in jobs]) and len(jobs) <= 8:
            preds = networks.multi_forward([(d, x) for d, x, _ in jobs])      # one launch per layer for all chains
            return self.criterionGAN.weighted_sum(preds, [r for _, _, r in jobs], weights)
        streams = self._streams[:len(jobs)] if self._streams else None
        losses = []
        if not streams:
            for netD, x, is_real in jobs:
                losses.append(self.criterionGAN(netD.forward(x), is_real))
        else:
            cur = torch.cuda.current_stream()
            # The same net may run on two side streams (fake and real batch): make its derived weight copies (one launch, keyed on the
            # host) on the CURRENT stream before the fork, so that no chain reads a copy another stream is still writing.
            for netD in {id(d): d for d, _, _ in jobs}.values():
                if hasattr(netD, "_refresh_derived"):
                    netD._refresh_derived()
            for st, (netD, x, is_real) in zip(streams, jobs):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    l = self.criterionGAN(netD.forward(x), is_real)
                l.record_stream(cur)
                losses.append(l)
            for st in streams:
                cur.wait_stream(st)
        total = 0
        for l, w in zip(losses, weights):
            total = total + l * w
        return total, torch.stack([l.detach() for l in losses])

    def _join_streams(self):
        """The chains' backward kernels (weight-gradient atomics included) ran on the side streams: the
        optimizer step that follows on the current stream must wait for them."""
        cur = torch.cuda.current_stream()
        for st in self._streams:
            cur.wait_stream(st)

    def backward_D(self):
        """loss_D = 0.5 * (sum_i BCE(D_i(fake), 0) + sum_i BCE(D_i(real), 1))   (fcgan_model.py:146-163)"""
        fake = self._pool_override if self._pool_override is not None else self.fake_pool.query(self.fake)
        fake = fake.detach()
        n = self.n_netD
        self.loss_D, self._each_D = self._d_losses([(d, fake, False) for d in self.netD] + [(d, self.real, True) for d in self.netD],
                                                   [0.5] * (2 * n))
        self._backward(self.loss_D)
        self._join_streams()

    def backward_G(self):
        """loss_G = sum_i lambda_i * BCE(D_i(fake), 1)  (log-D trick) or -sum_i lambda_i * BCE(D_i(fake), 0)
        (fcgan_model.py:165-176)"""
        skip = getattr(self.opt, 'skip_wasted_D_wgrad', False)
        for netD in self.netD:
            netD.compute_param_grads = not skip
        trick = not self.opt.no_logD_trick
        self.loss_G, self._each_G = self._d_losses([(d, self.fake, trick) for d in self.netD],
                                                   [l if trick else -l for l in self.opt.lambda_D])
        for netD in self.netD:
            netD.compute_param_grads = True
        self._backward(self.loss_G)
        self._join_streams()

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
        return OrderedDict([('G_GAN', float(self.loss_G.detach())), ('D_real', float(self.loss_D_real)),
                            ('D_fake', float(self.loss_D_fake))])

    def get_current_visuals(self, save_real=False, save_as_single_image=True):
        out = OrderedDict()
        if self.isTrain or save_real:
            out['real'] = self.real.detach()
        out['fake'] = self.fake.detach()
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
