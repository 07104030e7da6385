"""CPU oracle for the supervised-gan conv G/D training hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch fp32 *restatement* of the
reference's algorithm for the path BASELINE.json names.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product path (`supervised-gan_amd/`) never does and fails loudly without its HIP
library.

Parity status: PINNED.  `oracle/make_golden.py` imports the real reference from
/root/reference in the build container, runs it on numpy-seeded weights/inputs and
commits the results under tests/golden/; tests/test_oracle_golden.py checks every
function here against those vectors.

All arithmetic is fp32 on CPU, autograd supplies the backward exactly as it does in
the reference (the reference has no hand-written backward either).

Reference map (paths relative to /root/reference):
  matlab_style_gauss2D / init_gauss_filters   models/networks.py:22-40
  weights_init                                models/networks.py:13-19
  FCGANGenerator                              models/networks.py:493-540
  NLayerDiscriminator                         models/networks.py:798-847
  define_D gaussian init (py2 int division)   models/networks.py:124-129
  GANLoss                                     models/networks.py:152-185
  GANLossMultiClass (+ backward_D2_multiclass) models/networks.py:188-202, models/twostage_cycle_model.py:302-335
  WeightedL1Loss                              models/networks.py:205-214
  UnetGenerator / UnetSkipConnectionBlock     models/networks.py:318-419
  CGANModel of cgan2 (two label images)       models/cgan2_model.py:129-233
  CGANCycleModel                              models/cgan_cycle_model.py:129-240
  CGANCycleModel of cgan2_cycle               models/cgan2_cycle_model.py:114-262
  AutoEncoder                                 models/networks.py:421-490
  DCGANGenerator / DCGANDiscriminator         models/networks.py:1015-1129
  FCGANGeneratorStar                          models/networks.py:543-640
  SegmentationModel step recipe               models/segm_model.py:145-263, models/loss.py:6-12
  SegmentationCycleModel step recipe          models/segm_cycle_model.py:159-281
  CascadedRefinementNetwork / Crn*Block       models/networks.py:642-794
  CGANModel step recipe                       models/cgan_model.py:134-226
  TwoStageCycleModel step recipe              models/twostage_cycle_model.py:193-438
  FCGANModel step recipe                      models/fcgan_model.py:124-193
  Adam hyper-parameters                       models/fcgan_model.py:98-109, options/train_options.py:16-17
  ImagePool                                   util/image_pool.py:6-42
  LR schedule                                 models/fcgan_model.py:228-236
"""
from __future__ import annotations

import math
import random
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
IN_EPS = 1e-5


# ----------------------------------------------------------------------------------
# deterministic, platform-independent tensors (numpy legacy RandomState is stable)
# ----------------------------------------------------------------------------------
def np_normal(seed: int, shape, mean=0.0, std=1.0) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(size=tuple(shape)) * std + mean).astype(np.float32))


def np_uniform(seed: int, shape, lo=-1.0, hi=1.0) -> torch.Tensor:
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.uniform(lo, hi, size=tuple(shape)).astype(np.float32))


# ----------------------------------------------------------------------------------
# Gaussian pre-filter (models/networks.py:22-40)
# ----------------------------------------------------------------------------------
def gauss2d(kw: int, sigma: float) -> np.ndarray:
    """fspecial('gaussian') restated: models/networks.py:22-33."""
    m = (kw - 1.0) / 2.0
    y, x = np.ogrid[-m:m + 1, -m:m + 1]
    h = np.exp(-(x * x + y * y) / (2.0 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    s = h.sum()
    if s != 0:
        h /= s
    return h


def gauss_filter_weight(nc: int, scale_factor: int) -> torch.Tensor:
    """Dense block-diagonal [nc,nc,k,k] weight; sigma = scale_factor // 2 (the
    reference is Python-2 code: models/networks.py:127-129, :808-811)."""
    sigma = scale_factor // 2
    kw = 4 * sigma + 1
    w = np.zeros((nc, nc, kw, kw))
    g = gauss2d(kw, sigma)
    for i in range(nc):
        w[i, i] = g
    return torch.from_numpy(w.astype(np.float32))


# ----------------------------------------------------------------------------------
# parameter construction with the reference's key names / shapes / init distributions
# ----------------------------------------------------------------------------------
def fcgan_g_channels(ngf: int, n_layers: int):
    """Channel plan of FCGANGenerator (models/networks.py:499-530)."""
    mults = [min(2 ** (n_layers - 1), 8)]
    for n in range(1, n_layers):
        mults.append(min(2 ** (n_layers - n - 1), 8))
    return [ngf * m for m in mults]


def init_fcgan_g(seed: int, noise_nc: int, out_nc: int, ngf: int = 32, n_layers: int = 5,
                 use_dropout: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """state_dict of FCGANGenerator(use_fcn=True) with numpy-seeded values drawn from the
    reference's init distributions (weights_init: conv N(0,.02), BN gamma N(1,.02), beta 0;
    conv biases U(+-1/sqrt(fan_in)) = torch default for ConvTranspose2d: fan_in = Cout*k*k)."""
    ch = fcgan_g_channels(ngf, n_layers)
    sd = OrderedDict()
    s = seed * 1000
    cin = noise_nc
    idx = 0
    for li, cout in enumerate(ch):
        sd[f"model.{idx}.weight"] = np_normal(s, (cin, cout, 4, 4), 0.0, 0.02); s += 1
        if li > 0:
            bound = 1.0 / math.sqrt(cout * 16)
            sd[f"model.{idx}.bias"] = np_uniform(s, (cout,), -bound, bound); s += 1
        sd[f"model.{idx + 1}.weight"] = np_normal(s, (cout,), 1.0, 0.02); s += 1
        sd[f"model.{idx + 1}.bias"] = torch.zeros(cout)
        sd[f"model.{idx + 1}.running_mean"] = torch.zeros(cout)
        sd[f"model.{idx + 1}.running_var"] = torch.ones(cout)
        sd[f"model.{idx + 1}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        cin = cout
        idx += 4 if (use_dropout and li > 0) else 3      # blocks above the first carry an nn.Dropout module (networks.py:513-521)
    sd[f"model.{idx}.weight"] = np_normal(s, (cin, out_nc, 4, 4), 0.0, 0.02)
    return sd


def nlayer_d_plan(input_nc: int, ndf: int, n_layers: int, logit_nc: int = 1):
    """[(idx, cin, cout, stride, has_norm)] of NLayerDiscriminator.model convs
    (models/networks.py:814-835); final logits conv last (logit_nc = num_classes when num_classes > 2, :806)."""
    plan = [(0, input_nc, ndf, 2, False)]
    nf = 1
    idx = 2
    for n in range(1, n_layers):
        nf_prev, nf = nf, min(2 ** n, 8)
        plan.append((idx, ndf * nf_prev, ndf * nf, 2, True)); idx += 3
    nf_prev, nf = nf, min(2 ** n_layers, 8)
    plan.append((idx, ndf * nf_prev, ndf * nf, 1, True)); idx += 3
    plan.append((idx, ndf * nf, logit_nc, 1, False))
    return plan


def init_nlayer_d(seed: int, input_nc: int, ndf: int = 32, n_layers: int = 3, scale_factor: int = 1,
                  logit_nc: int = 1) -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict()
    s = seed * 1000
    if scale_factor > 1:
        sd["gauss_filter.0.weight"] = gauss_filter_weight(input_nc, scale_factor)
    for idx, cin, cout, _stride, _norm in nlayer_d_plan(input_nc, ndf, n_layers, logit_nc):
        sd[f"model.{idx}.weight"] = np_normal(s, (cout, cin, 4, 4), 0.0, 0.02); s += 1
        bound = 1.0 / math.sqrt(cin * 16)
        sd[f"model.{idx}.bias"] = np_uniform(s, (cout,), -bound, bound); s += 1
    return sd


# ----------------------------------------------------------------------------------
# network forwards (functional; autograd gives the backward)
# ----------------------------------------------------------------------------------
def fcgan_g_forward(sd, z, n_layers: int = 5, update_running: bool = True, tanh: bool = True,
                    taps: dict | None = None, use_fcn: bool = True, use_dropout: bool = False, mask_seed: int = 0):
    """FCGANGenerator.forward (models/networks.py:535-540), BatchNorm always in train mode
    (the reference never calls .eval()).  `taps` (optional dict) receives raw conv outputs."""
    x = z
    idx = 0
    for li in range(n_layers):
        first_1x1 = li == 0 and not use_fcn     # --noiseSize 1: ConvT k4 s1 p0 on the 1x1 latent (models/networks.py:503-504)
        x = F.conv_transpose2d(x, sd[f"model.{idx}.weight"], sd.get(f"model.{idx}.bias"), stride=1 if first_1x1 else 2,
                               padding=0 if first_1x1 else 1)
        if taps is not None:
            taps[f"conv{li}"] = x
        rm = sd[f"model.{idx + 1}.running_mean"] if update_running else None
        rv = sd[f"model.{idx + 1}.running_var"] if update_running else None
        x = F.batch_norm(x, rm, rv, sd[f"model.{idx + 1}.weight"], sd[f"model.{idx + 1}.bias"],
                         training=True, momentum=BN_MOMENTUM, eps=BN_EPS)
        if update_running and f"model.{idx + 1}.num_batches_tracked" in sd:
            sd[f"model.{idx + 1}.num_batches_tracked"] += 1
        if use_dropout and li > 0:      # ConvT -> BatchNorm -> Dropout(0.5) -> ReLU (:513-521); the li-th mask of the pass
            x = x * dropout_mask_np(mask_seed + li, x.shape)
        x = F.relu(x)
        idx += 4 if (use_dropout and li > 0) else 3
    x = F.conv_transpose2d(x, sd[f"model.{idx}.weight"], None, stride=2, padding=1)
    if taps is not None:
        taps[f"conv{n_layers}"] = x
    return torch.tanh(x) if tanh else x


def gauss_down(x, w, scale_factor: int):
    """gauss_filter = Conv2d(k=4s+1, pad=2s, no bias) then AvgPool2d(kernel 1, stride s)
    (models/networks.py:807-813)."""
    sigma = scale_factor // 2
    y = F.conv2d(x, w, None, stride=1, padding=2 * sigma)
    return F.avg_pool2d(y, kernel_size=1, stride=scale_factor)


def nlayer_d_forward(sd, x, n_layers: int = 3, scale_factor: int = 1, use_sigmoid: bool = True,
                     taps: dict | None = None):
    """NLayerDiscriminator.forward with InstanceNorm2d(affine=False) (models/networks.py:841-847)."""
    if scale_factor > 1:
        x = gauss_down(x, sd["gauss_filter.0.weight"], scale_factor)
    input_nc = x.shape[1]
    ndf = sd["model.0.weight"].shape[0]
    logit_nc = sd[f"model.{nlayer_d_plan(input_nc, ndf, n_layers)[-1][0]}.weight"].shape[0]
    for li, (idx, _cin, _cout, stride, has_norm) in enumerate(nlayer_d_plan(input_nc, ndf, n_layers, logit_nc)):
        x = F.conv2d(x, sd[f"model.{idx}.weight"], sd[f"model.{idx}.bias"], stride=stride, padding=2)
        if taps is not None:
            taps[f"conv{li}"] = x
        last = li == n_layers + 1
        if has_norm:
            x = F.instance_norm(x, eps=IN_EPS)
        if not last:
            x = F.leaky_relu(x, 0.2)
    return torch.sigmoid(x) if use_sigmoid else x


def init_nlayer_d_sep(seed: int, ndf: int = 8, n_layers: int = 3, scale_factor: int = 1) -> "OrderedDict[str, torch.Tensor]":
    """Numpy-seeded state_dict of NLayerDiscriminatorSep (models/networks.py:851-925), InstanceNorm: netA (2 -> ndf -> 2 ndf), netB
    (1 -> ndf -> 2 ndf), model (4 ndf -> ... -> 1)."""
    sd = OrderedDict()
    s = seed * 1000

    def conv(key, cout, cin):
        nonlocal s
        sd[key + ".weight"] = np_normal(s, (cout, cin, 4, 4), 0.0, 0.02); s += 1
        bound = 1.0 / math.sqrt(cin * 16)
        sd[key + ".bias"] = np_uniform(s, (cout,), -bound, bound); s += 1
    if scale_factor > 1:
        sd["gauss_filter.0.weight"] = gauss_filter_weight(3, scale_factor)
    for net, cin in (("netA", 2), ("netB", 1)):
        conv(f"{net}.0", ndf, cin)
        conv(f"{net}.2", 2 * ndf, ndf)
    nf, idx = 4, 0
    for n in range(2, n_layers):
        nf_prev, nf = nf, min(2 ** n, 8)
        conv(f"model.{idx}", ndf * nf, ndf * nf_prev); idx += 3
    nf_prev, nf = nf, min(2 ** n_layers, 8)
    conv(f"model.{idx}", ndf * nf, ndf * nf_prev); idx += 3
    conv(f"model.{idx}", 1, ndf * nf)
    return sd


def nlayer_d_sep_forward(sd, x, n_layers: int = 3, scale_factor: int = 1, use_sigmoid: bool = True):
    """NLayerDiscriminatorSep.forward to its evident intent (models/networks.py:927-942): the label channels through netA, the image
    channel through netB (the reference's CPU branch calls netA on both, :940, and raises; its data_parallel branch uses netB),
    concatenated, then `model`.  InstanceNorm2d(affine=False)."""
    if scale_factor > 1:
        x = gauss_down(x, sd["gauss_filter.0.weight"], scale_factor)
    ys = []
    for net, sl in (("netA", slice(0, 2)), ("netB", slice(2, 3))):
        y = F.leaky_relu(F.conv2d(x[:, sl], sd[f"{net}.0.weight"], sd[f"{net}.0.bias"], stride=2, padding=2), 0.2)
        y = F.conv2d(y, sd[f"{net}.2.weight"], sd[f"{net}.2.bias"], stride=2, padding=2)
        ys.append(F.leaky_relu(F.instance_norm(y, eps=IN_EPS), 0.2))
    y = torch.cat(ys, 1)
    idx = 0
    for n in range(2, n_layers):
        y = F.leaky_relu(F.instance_norm(F.conv2d(y, sd[f"model.{idx}.weight"], sd[f"model.{idx}.bias"], stride=2, padding=2), eps=IN_EPS), 0.2)
        idx += 3
    y = F.leaky_relu(F.instance_norm(F.conv2d(y, sd[f"model.{idx}.weight"], sd[f"model.{idx}.bias"], stride=1, padding=2), eps=IN_EPS), 0.2)
    idx += 3
    y = F.conv2d(y, sd[f"model.{idx}.weight"], sd[f"model.{idx}.bias"], stride=1, padding=2)
    return torch.sigmoid(y) if use_sigmoid else y


def gan_loss(pred, target_is_real: bool, use_lsgan: bool = False):
    """GANLoss.__call__ (models/networks.py:183-185): BCELoss / MSELoss vs a constant map."""
    t = torch.full_like(pred, 1.0 if target_is_real else 0.0)
    return F.mse_loss(pred, t) if use_lsgan else F.binary_cross_entropy(pred, t)


def gan_loss_multiclass(pred, label: int):
    """GANLossMultiClass.__call__ (models/networks.py:188-202): CrossEntropyLoss over the class channel of every pixel."""
    nc = pred.shape[1]
    flat = pred.permute(0, 2, 3, 1).contiguous().view(-1, nc)
    return F.cross_entropy(flat, torch.full((flat.shape[0],), int(label), dtype=torch.long))


def weighted_l1(x, y, w=None):
    """WeightedL1Loss (models/networks.py:209-214)."""
    z = torch.abs(x - y)
    if w is not None:
        z = z * w
    return z.mean()


# ----------------------------------------------------------------------------------
# Adam (torch.optim.Adam default form) restated with explicit tensor ops
# ----------------------------------------------------------------------------------
class Adam:
    def __init__(self, params, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        step_size = self.lr / bc1
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-step_size)


# ----------------------------------------------------------------------------------
# ImagePool (util/image_pool.py:6-33), python `random` driven like the reference
# ----------------------------------------------------------------------------------
class ImagePool:
    def __init__(self, pool_size=50, reject=0.5):
        self.pool_size, self.reject = pool_size, reject
        self.num_imgs, self.images = 0, []

    def query(self, images):
        if self.pool_size == 0:
            return images
        out = []
        for image in images.detach():
            image = image.unsqueeze(0)
            if self.num_imgs < self.pool_size:
                self.num_imgs += 1
                self.images.append(image)
                out.append(image)
            else:
                p = random.uniform(0, 1)
                if p > self.reject:
                    rid = random.randint(0, self.pool_size - 1)
                    tmp = self.images[rid].clone()
                    self.images[rid] = image
                    out.append(tmp)
                else:
                    out.append(image)
        return torch.cat(out, 0)


# ----------------------------------------------------------------------------------
# the fcgan training step (models/fcgan_model.py:124-193)
# ----------------------------------------------------------------------------------
class FCGANConfig:
    """README fcgan flags (README.md:33)."""
    def __init__(self, input_nc=2, ngf=32, ndf=32, n_layers_G=5, n_layers_D=(3, 3, 3), scale_factor=(1, 2, 4),
                 lambda_D=(0.5, 0.4, 0.1), noise_nc=8, noiseSize=8, n_update_D=1, n_update_G=2,
                 lr=2e-4, beta1=0.5, pool_size=50, no_lsgan=True, no_logD_trick=False, batchSize=1):
        self.__dict__.update(locals())
        del self.__dict__["self"]

    @property
    def fineSize(self):
        return self.noiseSize * 2 ** (self.n_layers_G + 1)


class FCGANOracle:
    """FCGANModel restated (initialize :32-116, forward :124-128, backward_D :146-163,
    backward_G :165-176, optimize_parameters :178-193)."""

    def __init__(self, cfg: FCGANConfig, seed: int = 0):
        self.cfg = cfg
        self.G = init_fcgan_g(seed + 1, cfg.noise_nc, cfg.input_nc, cfg.ngf, cfg.n_layers_G)
        self.D = [init_nlayer_d(seed + 2 + i, cfg.input_nc, cfg.ndf, nl, sf)
                  for i, (nl, sf) in enumerate(zip(cfg.n_layers_D, cfg.scale_factor))]
        for k, v in self.G.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
        for d in self.D:
            for k, v in d.items():
                if v.is_floating_point():
                    v.requires_grad_(True)      # gauss_filter weights require grad in the reference too
        g_params = [v for k, v in self.G.items() if v.requires_grad]
        d_params = [v for d in self.D for k, v in d.items() if k.startswith("model.")]
        self.opt_G = Adam(g_params, cfg.lr, cfg.beta1)
        self.opt_D = Adam(d_params, cfg.lr, cfg.beta1)
        self.pool = ImagePool(cfg.pool_size)
        self.noise_iter = None  # set by caller: iterator yielding z tensors

    def _z(self):
        return next(self.noise_iter)

    def forward(self):
        self.noise = self._z()
        self.fake = fcgan_g_forward(self.G, self.noise, self.cfg.n_layers_G)

    def _d(self, i, x):
        c = self.cfg
        return nlayer_d_forward(self.D[i], x, c.n_layers_D[i], c.scale_factor[i], use_sigmoid=c.no_lsgan)

    def backward_D(self):
        c = self.cfg
        fake = self.pool.query(self.fake)
        self.loss_D_fake = sum(gan_loss(self._d(i, fake.detach()), False, not c.no_lsgan) for i in range(len(self.D)))
        self.loss_D_real = sum(gan_loss(self._d(i, self.real), True, not c.no_lsgan) for i in range(len(self.D)))
        self.loss_D = (self.loss_D_fake + self.loss_D_real) * 0.5
        self.loss_D.backward()

    def backward_G(self):
        c = self.cfg
        loss = 0
        for i, lam in enumerate(c.lambda_D):
            pred = self._d(i, self.fake)
            if not c.no_logD_trick:
                loss = loss + gan_loss(pred, True, not c.no_lsgan) * lam
            else:
                loss = loss - gan_loss(pred, False, not c.no_lsgan) * lam
        self.loss_G = loss
        self.loss_G.backward()

    def _zero_all_D_grads(self):
        for d in self.D:
            for v in d.values():
                v.grad = None

    def optimize_parameters(self, real):
        c = self.cfg
        self.real = real
        self.forward()
        for _ in range(c.n_update_D):
            self.opt_D.zero_grad()
            self.backward_D()
            self.opt_D.step()
            if c.n_update_D > 1:
                self.forward()
        for _ in range(c.n_update_G):
            self.opt_G.zero_grad()
            self.backward_G()
            self.opt_G.step()
            if c.n_update_G > 1:
                self.forward()

    def step1_with_captures(self, real):
        """First training step in optimize_parameters' order with the pre-Adam quantities captured
        (same capture points as oracle/make_golden.py::golden_step)."""
        c = self.cfg
        assert c.n_update_D == 1
        cap = {}
        self.real = real
        self.forward()
        cap["fake"] = self.fake.detach().clone()
        self.opt_D.zero_grad()
        self.backward_D()
        cap["gradD"] = [{k: v.grad.detach().clone() for k, v in d.items() if k.startswith("model.")} for d in self.D]
        cap["loss_D"] = [float(self.loss_D_real.detach()), float(self.loss_D_fake.detach())]
        self.opt_D.step()
        for it in range(c.n_update_G):
            self.opt_G.zero_grad()
            self.backward_G()
            if it == 0:
                cap["gradG"] = {k: v.grad.detach().clone() for k, v in self.G.items() if v.grad is not None}
                cap["loss_G"] = float(self.loss_G.detach())
            self.opt_G.step()
            if c.n_update_G > 1:
                self.forward()
        return cap

    def probe_G(self, real):
        """forward() + backward_G() on the initial weights (capture point 'probeG' of make_golden.py)."""
        self.real = real
        self.forward()
        self.opt_G.zero_grad()
        self.opt_D.zero_grad()
        self.backward_G()
        return {"gradG": {k: v.grad.detach().clone() for k, v in self.G.items() if v.grad is not None},
                "gradD": [{k: v.grad.detach().clone() for k, v in d.items() if k.startswith("model.")} for d in self.D],
                "loss_G": float(self.loss_G.detach())}

    def losses(self):
        return {"G_GAN": float(self.loss_G.detach()), "D_real": float(self.loss_D_real.detach()),
                "D_fake": float(self.loss_D_fake.detach())}


# ----------------------------------------------------------------------------------
# U-Net generator (models/networks.py:318-419)
# ----------------------------------------------------------------------------------
def unet_plan(num_downs: int, ngf: int, input_nc: int, output_nc: int, num_skips: int = -1, use_dropout: bool = False):
    """The convs of UnetGenerator as a list of levels.  Level l = 0..n-1: `down` conv producing x_l
    (c_l channels at H/2^(l+1)), `up` transposed conv of the block that wraps x_l.  Level 0 is the bare
    downconv/upconv pair of UnetGenerator.model (:356-358); levels 1..n-1 are UnetSkipConnectionBlocks,
    n-1 the innermost one (:330-354).  `skip[l]`: block l returns cat([y, x]) (:419)."""
    n = num_downs
    if num_skips < 0:
        num_skips = n
    c = [ngf * min(2 ** l, 8) for l in range(n)]
    skip = [False] + [num_skips >= n - l for l in range(1, n)]
    levels = []
    for l in range(n):
        prefix = "model.1" + ".model.3" * (l - 1) if l >= 1 else None
        innermost = l == n - 1
        if l == 0:
            down_key, up_key = "model.0", "model.3"
            up_in = c[0] * (2 if skip[1] else 1)
            lv = dict(down=(down_key, input_nc, c[0]), up=(up_key, up_in, output_nc), down_norm=False, up_norm=False)
        else:
            down_key = prefix + ".model.1"
            up_key = prefix + (".model.3" if innermost else ".model.5")
            up_in = c[l] if innermost else c[l] * (2 if skip[l + 1] else 1)
            lv = dict(down=(down_key, c[l - 1], c[l]), up=(up_key, up_in, c[l - 1]), down_norm=not innermost, up_norm=True)
        lv["dropout"] = bool(use_dropout and 4 <= l <= n - 2)      # the `num_downs - 5` ngf*8 blocks (:333-339)
        lv["skip"] = skip[l]
        lv["innermost"] = innermost
        levels.append(lv)
    return levels


def _next_key(key: str) -> str:
    """The next numbered child of the same nn.Sequential (a conv's norm layer follows it)."""
    parts = key.split(".")
    return ".".join(parts[:-1] + [str(int(parts[-1]) + 1)])


def _init_bn(sd, key: str, c: int, seed: int):
    """BatchNorm2d(affine=True) entries as weights_init leaves them (:17-19: gamma N(1, .02), beta 0) + fresh running statistics."""
    sd[key + ".weight"] = np_normal(seed, (c,), 1.0, 0.02)
    sd[key + ".bias"] = torch.zeros(c)
    sd[key + ".running_mean"] = torch.zeros(c)
    sd[key + ".running_var"] = torch.ones(c)
    sd[key + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def init_unet(seed: int, num_downs: int, input_nc: int, output_nc: int, ngf: int = 64, num_skips: int = -1, norm: str = "instance"):
    """numpy-seeded state dict with UnetGenerator's keys/shapes; N(0, 0.02) conv weights (weights_init
    :13-19), torch-default uniform biases; norm 'batch': the BatchNorm2d layers behind the down / up convs (:387-389)."""
    sd = OrderedDict()
    k = 0
    for lv in unet_plan(num_downs, ngf, input_nc, output_nc, num_skips):
        if norm == "batch":
            if lv["down_norm"]:
                _init_bn(sd, _next_key(lv["down"][0]), lv["down"][2], seed * 1000 + 500 + k)
            if lv["up_norm"]:
                _init_bn(sd, _next_key(lv["up"][0]), lv["up"][2], seed * 1000 + 501 + k)
        key, ci, co = lv["down"]
        sd[key + ".weight"] = np_normal(seed * 1000 + k, (co, ci, 4, 4), 0.0, 0.02)
        b = 1.0 / math.sqrt(ci * 16)
        sd[key + ".bias"] = np_uniform(seed * 1000 + k + 1, (co,), -b, b)
        key, ci, co = lv["up"]
        sd[key + ".weight"] = np_normal(seed * 1000 + k + 2, (ci, co, 4, 4), 0.0, 0.02)
        b = 1.0 / math.sqrt(co * 16)            # torch computes fan_in from weight.size(1) for ConvTranspose2d
        sd[key + ".bias"] = np_uniform(seed * 1000 + k + 3, (co,), -b, b)
        k += 4
    return sd


def init_resnet(seed: int, input_nc: int, output_nc: int, ngf: int, n_blocks: int, use_dropout: bool = False, norm: str = "instance"):
    """numpy-seeded state dict with ResnetGenerator's keys / shapes (models/networks.py:232-262, :282-300, norm 'instance'):
    N(0, 0.02) conv weights (weights_init :13-19), torch-default uniform biases."""
    sd = OrderedDict()
    C = 4 * ngf
    second = 6 if use_dropout else 5
    convs = [("model.1", False, input_nc, ngf, 7), ("model.4", False, ngf, 2 * ngf, 3), ("model.7", False, 2 * ngf, C, 3)]
    for i in range(n_blocks):
        convs += [(f"model.{10 + i}.conv_block.1", False, C, C, 3), (f"model.{10 + i}.conv_block.{second}", False, C, C, 3)]
    nb = 10 + n_blocks
    convs += [(f"model.{nb}", True, C, 2 * ngf, 3), (f"model.{nb + 3}", True, 2 * ngf, ngf, 3), (f"model.{nb + 7}", False, ngf, output_nc, 7)]
    for k, (key, tr, ci, co, ks) in enumerate(convs):
        sd[key + ".weight"] = np_normal(seed * 1000 + 2 * k, (ci, co, ks, ks) if tr else (co, ci, ks, ks), 0.0, 0.02)
        b = 1.0 / math.sqrt((co if tr else ci) * ks * ks)      # torch computes fan_in from weight.size(1)
        sd[key + ".bias"] = np_uniform(seed * 1000 + 2 * k + 1, (co,), -b, b)
        if norm == "batch" and k < len(convs) - 1:      # every conv but the last is followed by norm_layer(co)
            _init_bn(sd, _next_key(key), co, seed * 1000 + 700 + k)
    return sd


def resnet_forward(sd, x, n_blocks: int, use_dropout: bool = False, mask_seed: int = 0, tanh: bool = True, use_residual: bool = False,
                   norm: str = "instance"):
    """ResnetGenerator.forward (models/networks.py:221-268) with ResnetBlock (:271-311), padding_type 'reflect', InstanceNorm:
    the i-th block's Dropout(0.5) mask is dropout_mask_np(mask_seed + i, shape)."""
    def inorm(t, conv_key):
        """norm_layer behind the conv `conv_key`: InstanceNorm2d(affine=False), or BatchNorm2d in train mode with --norm batch."""
        if norm != "batch":
            return F.instance_norm(t, eps=1e-5)
        k = _next_key(conv_key)
        sd[k + ".num_batches_tracked"] += 1
        return F.batch_norm(t, sd[k + ".running_mean"], sd[k + ".running_var"], sd[k + ".weight"], sd[k + ".bias"], training=True,
                            momentum=BN_MOMENTUM, eps=BN_EPS)

    rpad = lambda t, p: F.pad(t, (p, p, p, p), mode="reflect")      # noqa: E731
    h = F.relu(inorm(F.conv2d(rpad(x, 3), sd["model.1.weight"], sd["model.1.bias"]), "model.1"))
    for key in ("model.4", "model.7"):
        h = F.relu(inorm(F.conv2d(h, sd[key + ".weight"], sd[key + ".bias"], stride=2, padding=1), key))
    second = 6 if use_dropout else 5
    for i in range(n_blocks):
        p = f"model.{10 + i}.conv_block."
        t = F.relu(inorm(F.conv2d(rpad(h, 1), sd[p + "1.weight"], sd[p + "1.bias"]), p + "1"))
        if use_dropout:
            t = t * dropout_mask_np(mask_seed + i, t.shape)
        h = h + inorm(F.conv2d(rpad(t, 1), sd[p + f"{second}.weight"], sd[p + f"{second}.bias"]), p + f"{second}")
    nb = 10 + n_blocks
    for key in (f"model.{nb}", f"model.{nb + 3}"):
        h = F.relu(inorm(F.conv_transpose2d(h, sd[key + ".weight"], sd[key + ".bias"], stride=2, padding=1, output_padding=1), key))
    y = F.conv2d(rpad(h, 3), sd[f"model.{nb + 7}.weight"], sd[f"model.{nb + 7}.bias"])
    # the reference applies Tanh TWICE without --use_residual: once as the last module of self.model (:261-262) and again in
    # forward() (:268: `nn.Tanh()(y)`); with --use_residual the Sequential ends in the conv (:258-259) and forward() is tanh(x + y)
    if use_residual:
        return torch.tanh(x + y) if tanh else x + y
    return torch.tanh(torch.tanh(y)) if tanh else y


def dropout_mask_np(seed: int, shape, p: float = 0.5) -> torch.Tensor:
    """Deterministic Dropout(p) keep-mask (0 or 1 / (1 - p); 0 or 2 for the default 0.5), keyed on the tensor shape
    (make_golden.py injects the same masks into the reference)."""
    rs = np.random.RandomState(seed + int(shape[1]) * 7 + int(shape[2]))
    return torch.from_numpy((rs.uniform(size=tuple(shape)) >= p).astype(np.float32) * np.float32(1.0 / (1.0 - p)))


def gauss_noise_np(seed: int, shape) -> torch.Tensor:
    return np_normal(seed + int(shape[1]) * 131 + int(shape[2]), shape)


def unet_forward(sd, x, num_downs: int, ngf: int, num_skips: int = -1, use_dropout: bool = False, mask_seed: int = 0,
                 add_gaussian_noise: bool = False, gaussian_sigma: float = 0.1, noise_seed: int = 0, tanh: bool = True,
                 use_residual: bool = False, norm: str = "instance"):
    """UnetGenerator.forward (:362-367: `activation(x + y) if self.use_residual else activation(y)`) with
    UnetSkipConnectionBlock.forward (:409-419) inlined."""
    input_nc = sd["model.0.weight"].shape[1]
    output_nc = sd["model.3.weight"].shape[1]
    levels = unet_plan(num_downs, ngf, input_nc, output_nc, num_skips, use_dropout)

    def nrm(h, conv_key):
        """norm_layer behind the conv `conv_key` (get_norm_layer, :43-50): InstanceNorm2d(affine=False) or BatchNorm2d in train mode."""
        if norm != "batch":
            return F.instance_norm(h, eps=IN_EPS)
        k = _next_key(conv_key)
        sd[k + ".num_batches_tracked"] += 1
        return F.batch_norm(h, sd[k + ".running_mean"], sd[k + ".running_var"], sd[k + ".weight"], sd[k + ".bias"], training=True,
                            momentum=BN_MOMENTUM, eps=BN_EPS)

    def block(l, xin):
        lv = levels[l]
        dk, uk = lv["down"][0], lv["up"][0]
        h = F.leaky_relu(xin, 0.2)
        h = F.conv2d(h, sd[dk + ".weight"], sd[dk + ".bias"], stride=2, padding=1)
        if not lv["innermost"]:
            h = nrm(h, dk)
            h = block(l + 1, h)
        h = F.relu(h)
        h = F.conv_transpose2d(h, sd[uk + ".weight"], sd[uk + ".bias"], stride=2, padding=1)
        h = nrm(h, uk)
        if lv["dropout"]:
            h = h * dropout_mask_np(mask_seed, h.shape)
        if add_gaussian_noise:
            h = h + gaussian_sigma * gauss_noise_np(noise_seed, h.shape)
        return torch.cat([h, xin], 1) if lv["skip"] else h

    h = F.conv2d(x, sd["model.0.weight"], sd["model.0.bias"], stride=2, padding=1)
    h = block(1, h)
    h = F.relu(h)
    h = F.conv_transpose2d(h, sd["model.3.weight"], sd["model.3.bias"], stride=2, padding=1)
    if use_residual:
        h = x + h
    return torch.tanh(h) if tanh else h


def norm_cancelled_keys_unet(num_downs: int, ngf: int = 64, num_skips: int = -1):
    """Conv biases of the U-Net that feed an InstanceNorm (analytically zero gradient, see
    norm_cancelled_keys_g)."""
    keys = set()
    for lv in unet_plan(num_downs, ngf, 1, 1, num_skips):
        if lv["down_norm"]:
            keys.add(lv["down"][0] + ".bias")
        if lv["up_norm"]:
            keys.add(lv["up"][0] + ".bias")
    return keys


# ----------------------------------------------------------------------------------
# AutoEncoder generator (models/networks.py:421-490)
# ----------------------------------------------------------------------------------
def autoencoder_plan(input_nc: int, output_nc: int, n_layers: int, ngf: int, use_dropout: bool = False):
    """[(sequential index, kind, cin, cout, bias, normed, dropout p)]; with use_dropout every block but the first of each half is
    conv, norm, Dropout, ReLU (four modules: networks.py:441-447 p = 0.2, :468-474 p = 0.5)."""
    step = 4 if use_dropout else 3
    plan, idx, nf = [], 0, 1
    plan.append((idx, "conv", input_nc, ngf, True, True, 0.0))
    idx += 3
    for n in range(1, n_layers):
        nf_prev, nf = nf, min(2 ** n, 8)
        plan.append((idx, "conv", nf_prev * ngf, ngf * nf, True, True, 0.2 if use_dropout else 0.0))
        idx += step
    latent = min(2 ** n_layers, 8)
    plan.append((idx, "conv", nf * ngf, latent, False, False, 0.0))
    idx += 1
    nf = min(2 ** (n_layers - 1), 8)
    plan.append((idx, "convt", latent, ngf * nf, False, True, 0.0))
    idx += 3
    for n in range(1, n_layers):
        nf_prev, nf = nf, min(2 ** (n_layers - n - 1), 8)
        plan.append((idx, "convt", ngf * nf_prev, ngf * nf, True, True, 0.5 if use_dropout else 0.0))
        idx += step
    plan.append((idx, "convt", ngf, output_nc, False, False, 0.0))
    return plan


def init_autoencoder(seed: int, input_nc: int, output_nc: int, n_layers: int = 3, ngf: int = 64, use_dropout: bool = False):
    sd = OrderedDict()
    for k, (idx, kind, ci, co, bias, _n, _p) in enumerate(autoencoder_plan(input_nc, output_nc, n_layers, ngf, use_dropout)):
        shape = (co, ci, 4, 4) if kind == "conv" else (ci, co, 4, 4)
        sd[f"model.{idx}.weight"] = np_normal(seed * 1000 + 2 * k, shape, 0.0, 0.02)
        if bias:
            b = 1.0 / math.sqrt((ci if kind == "conv" else co) * 16)
            sd[f"model.{idx}.bias"] = np_uniform(seed * 1000 + 2 * k + 1, (co,), -b, b)
    return sd


def autoencoder_forward(sd, x, n_layers: int, ngf: int, use_dropout: bool = False, mask_seed=None):
    """mask_seed: the i-th dropout layer (in forward order) multiplies by dropout_mask_np(mask_seed + i, shape, p) -- training
    mode with injected masks; None: no dropout (eval mode, or a net built without it)."""
    input_nc, output_nc = sd["model.0.weight"].shape[1], None
    last = max(int(k.split(".")[1]) for k in sd)
    output_nc = sd[f"model.{last}.weight"].shape[1]
    h, nd = x, 0
    for idx, kind, ci, co, bias, normed, p in autoencoder_plan(input_nc, output_nc, n_layers, ngf, use_dropout):
        w, b = sd[f"model.{idx}.weight"], sd.get(f"model.{idx}.bias")
        h = F.conv2d(h, w, b, stride=2, padding=1) if kind == "conv" else F.conv_transpose2d(h, w, b, stride=2, padding=1)
        if normed:
            h = F.instance_norm(h, eps=IN_EPS)
            if p > 0 and mask_seed is not None:
                h = h * dropout_mask_np(mask_seed + nd, h.shape, p)
                nd += 1
            h = F.relu(h)
    return torch.tanh(h)


# ----------------------------------------------------------------------------------
# DCGAN generator / discriminator (models/networks.py:1015-1129): bias-free chains with BatchNorm
# ----------------------------------------------------------------------------------
def dcgan_g_plan(nz: int, nc: int, ngf: int):
    """[(sequential index, cin, cout, stride, pad, has_bn)]"""
    ch = [ngf * 8, ngf * 4, ngf * 2, ngf, int(ngf / 2)]
    plan = [(0, nz, ch[0], 1, 0, True)]
    for i in range(1, 5):
        plan.append((3 * i, ch[i - 1], ch[i], 2, 1, True))
    plan.append((15, ch[4], nc, 2, 1, False))
    return plan


def dcgan_d_plan(nc: int, ndf: int):
    ch = [int(ndf / 2), ndf, ndf * 2, ndf * 4, ndf * 8]
    plan = [(0, nc, ch[0], 2, 1, False)]
    for i in range(1, 5):
        plan.append((3 * i - 1, ch[i - 1], ch[i], 2, 1, True))
    plan.append((14, ch[4], 1, 1, 0, False))
    return plan


def _init_dcgan(seed, plan, transposed):
    sd = OrderedDict()
    for k, (idx, ci, co, _s, _p, bn) in enumerate(plan):
        sd[f"model.{idx}.weight"] = np_normal(seed * 1000 + 3 * k, (ci, co, 4, 4) if transposed else (co, ci, 4, 4), 0.0, 0.02)
        if bn:
            sd[f"model.{idx + 1}.weight"] = np_normal(seed * 1000 + 3 * k + 1, (co,), 1.0, 0.02)
            sd[f"model.{idx + 1}.bias"] = torch.zeros(co)
            sd[f"model.{idx + 1}.running_mean"] = torch.zeros(co)
            sd[f"model.{idx + 1}.running_var"] = torch.ones(co)
            sd[f"model.{idx + 1}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    return sd


def init_dcgan_g(seed: int, nz: int, nc: int, ngf: int):
    return _init_dcgan(seed, dcgan_g_plan(nz, nc, ngf), True)


def init_dcgan_d(seed: int, nc: int, ndf: int):
    return _init_dcgan(seed, dcgan_d_plan(nc, ndf), False)


def _bn_train(x, sd, idx):
    return F.batch_norm(x, sd[f"model.{idx}.running_mean"], sd[f"model.{idx}.running_var"], sd[f"model.{idx}.weight"],
                        sd[f"model.{idx}.bias"], training=True, momentum=0.1, eps=BN_EPS)


def dcgan_g_forward(sd, z, nz: int, nc: int, ngf: int):
    h = z
    for idx, _ci, _co, s_, p_, bn in dcgan_g_plan(nz, nc, ngf):
        h = F.conv_transpose2d(h, sd[f"model.{idx}.weight"], None, stride=s_, padding=p_)
        if bn:
            h = F.relu(_bn_train(h, sd, idx + 1))
    return torch.tanh(h)


def dcgan_d_forward(sd, x, nc: int, ndf: int):
    h = x
    plan = dcgan_d_plan(nc, ndf)
    for li, (idx, _ci, _co, s_, p_, bn) in enumerate(plan):
        h = F.conv2d(h, sd[f"model.{idx}.weight"], None, stride=s_, padding=p_)
        if bn:
            h = _bn_train(h, sd, idx + 1)
        if li < len(plan) - 1:
            h = F.leaky_relu(h, 0.2)
    return torch.sigmoid(h).view(-1, 1).squeeze(1)


# ----------------------------------------------------------------------------------
# FCGANGeneratorStar (models/networks.py:543-640): two bias-free deconv chains; chain b reads cat(a, b) of the level below
# ----------------------------------------------------------------------------------
def fcgan_star_plan(noise_nc: int, ngf: int):
    """[(module name, cin, cout, has_bn)] in the reference's module order: conv0a..conv5a, then conv0b..conv5b (:556-623)."""
    half = int(noise_nc / 2)
    ch = [ngf * 8, ngf * 8, ngf * 4, ngf * 2, ngf]
    plan = [("conv0a", half, ch[0], True)]
    for i in range(1, 5):
        plan.append((f"conv{i}a", ch[i - 1], ch[i], True))
    plan.append(("conv5a", ch[4], 1, False))
    plan.append(("conv0b", half, ch[0], True))
    for i in range(1, 5):
        plan.append((f"conv{i}b", 2 * ch[i - 1], ch[i], True))
    plan.append(("conv5b", 2 * ch[4], 1, False))
    return plan


def init_fcgan_star(seed: int, noise_nc: int, ngf: int):
    sd = OrderedDict()
    for k, (name, ci, co, bn) in enumerate(fcgan_star_plan(noise_nc, ngf)):
        sd[f"{name}.0.weight"] = np_normal(seed * 1000 + 2 * k, (ci, co, 4, 4), 0.0, 0.02)
        if bn:
            sd[f"{name}.1.weight"] = np_normal(seed * 1000 + 2 * k + 1, (co,), 1.0, 0.02)
            sd[f"{name}.1.bias"] = torch.zeros(co)
            sd[f"{name}.1.running_mean"] = torch.zeros(co)
            sd[f"{name}.1.running_var"] = torch.ones(co)
            sd[f"{name}.1.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    return sd


def fcgan_star_forward(sd, z, noise_nc: int):
    """FCGANGeneratorStar.forward (:625-640): noise1 = first half of z feeds chain b, noise2 = second half chain a; every
    b layer above the first reads cat([ha, hb]); output tanh(cat([ha, hb]))."""
    half = int(noise_nc / 2)

    def layer(name, h, bn=True):
        h = F.conv_transpose2d(h, sd[f"{name}.0.weight"], None, stride=2, padding=1)
        if not bn:
            return h
        h = F.batch_norm(h, sd[f"{name}.1.running_mean"], sd[f"{name}.1.running_var"], sd[f"{name}.1.weight"], sd[f"{name}.1.bias"],
                         training=True, momentum=BN_MOMENTUM, eps=BN_EPS)
        sd[f"{name}.1.num_batches_tracked"] += 1
        return F.relu(h)

    hb = layer("conv0b", z.narrow(1, 0, half))
    ha = layer("conv0a", z.narrow(1, half, half))
    for i in range(1, 6):
        hb = layer(f"conv{i}b", torch.cat([ha, hb], 1), bn=i < 5)
        ha = layer(f"conv{i}a", ha, bn=i < 5)
    return torch.tanh(torch.cat([ha, hb], 1))


# ----------------------------------------------------------------------------------
# Cascaded refinement network (models/networks.py:642-794)
# ----------------------------------------------------------------------------------
def crn_plan(input_nc: int, output_nc: int, noise_nc: int, ngf: int, upsample_mode: str = "convt", n_layers_block: int = 1,
             share_label_weights: bool = True):
    """[(key, kind, cin, cout, bias)] of every conv of CascadedRefinementNetwork in the reference's module order
    (blockh5 .. blockh0, then the label block(s)); kind 'conv3' = Conv2d(k3,s1,p1), 'convt' = ConvTranspose2d(k4,s2,p1)."""
    convs = []
    for s in range(5, -1, -1):
        cin = noise_nc + input_nc if s == 5 else 2 * ngf
        if upsample_mode == "convt":
            convs.append((f"blockh{s}.0.model.0", "convt", cin, ngf, False))            # :736-740
        elif upsample_mode == "bilinear":
            convs.append((f"blockh{s}.0.model.0", "conv3", cin, ngf, True))             # :741-746
        else:
            raise NotImplementedError(upsample_mode)
        for i in range(1, n_layers_block):                                              # CrnInterBlock :760-779
            convs.append((f"blockh{s}.1.model.{3 * (i - 1) + 1}", "conv3", ngf, ngf, True))
        convs.append((f"blockh{s}.1.model.{3 * (n_layers_block - 1) + 1}", "conv3", ngf, output_nc if s == 0 else ngf, True))
    if share_label_weights:
        convs.append(("blockl.0", "conv3", input_nc, ngf, True))                        # :684-688
    else:
        for s in range(4, -1, -1):
            convs.append((f"blockl{s}.0", "conv3", input_nc, ngf, True))
    return convs


def _crn_norm_key(key: str, upsample_mode: str) -> str:
    """The norm_layer module behind a CRN conv: the next child, except in the bilinear upsample block (Conv2d, Upsample, norm: :749-753)."""
    if key.endswith(".0.model.0") and upsample_mode == "bilinear":
        return key[:-1] + "2"
    return _next_key(key)


def init_crn(seed: int, input_nc: int, output_nc: int, noise_nc: int, ngf: int = 64, upsample_mode: str = "convt",
             n_layers_block: int = 1, share_label_weights: bool = True, norm: str = "instance"):
    sd = OrderedDict()
    last = f"blockh0.1.model.{3 * (n_layers_block - 1) + 1}"
    for k, (key, kind, cin, cout, bias) in enumerate(crn_plan(input_nc, output_nc, noise_nc, ngf, upsample_mode, n_layers_block,
                                                              share_label_weights)):
        if kind == "convt":
            sd[key + ".weight"] = np_normal(seed * 1000 + 2 * k, (cin, cout, 4, 4), 0.0, 0.02)
        else:
            sd[key + ".weight"] = np_normal(seed * 1000 + 2 * k, (cout, cin, 3, 3), 0.0, 0.02)
            if bias:
                b = 1.0 / math.sqrt(cin * 9)
                sd[key + ".bias"] = np_uniform(seed * 1000 + 2 * k + 1, (cout,), -b, b)
        if norm == "batch" and key != last:
            _init_bn(sd, _crn_norm_key(key, upsample_mode), cout, seed * 1000 + 800 + k)
    return sd


def crn_forward(sd, label, noise, ngf: int, upsample_mode: str = "convt", n_layers_block: int = 1,
                share_label_weights: bool = True, tanh: bool = True, gauss_seed=None, gauss_sigma: float = 0.1, norm: str = "instance"):
    """CascadedRefinementNetwork.forward (:708-733) with CrnUpsampleBlock (:737-757) and CrnInterBlock (:760-787)
    inlined; InstanceNorm2d(affine=False), or with norm 'batch' BatchNorm2d in train mode (the shared label block's one is applied,
    and its running statistics updated, once per scale).  gauss_seed: --add_gaussian_noise, every upsample block but blockh0 adds
    gauss_sigma * gauss_noise_np(gauss_seed, shape) to its normalised output (:655-680,757-760)."""
    def conv3(x, key):
        return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=1, padding=1)

    def nrm(h, conv_key):
        if norm != "batch":
            return F.instance_norm(h, eps=IN_EPS)
        k = _crn_norm_key(conv_key, upsample_mode)
        sd[k + ".num_batches_tracked"] += 1
        return F.batch_norm(h, sd[k + ".running_mean"], sd[k + ".running_var"], sd[k + ".weight"], sd[k + ".bias"], training=True,
                            momentum=BN_MOMENTUM, eps=BN_EPS)

    def block(s, x):
        k0 = f"blockh{s}.0.model.0"
        if upsample_mode == "convt":
            h = F.conv_transpose2d(x, sd[k0 + ".weight"], None, stride=2, padding=1)
        else:
            h = F.interpolate(conv3(x, k0), scale_factor=2, mode="bilinear", align_corners=False)
        h = nrm(h, k0)
        if gauss_seed is not None and s > 0:
            h = h + gauss_sigma * gauss_noise_np(gauss_seed, h.shape)
        for i in range(n_layers_block):
            ki = f"blockh{s}.1.model.{3 * i + 1}"
            h = conv3(F.relu(h), ki)
            if not (s == 0 and i == n_layers_block - 1):
                h = nrm(h, ki)
        return h

    h = block(5, torch.cat([F.avg_pool2d(label, 64, 64), noise], 1))
    for s in range(4, -1, -1):
        l = F.avg_pool2d(label, 2 ** (s + 1), 2 ** (s + 1))
        kl = "blockl.0" if share_label_weights else f"blockl{s}.0"
        l = nrm(conv3(l, kl), kl)
        h = block(s, torch.cat([l, h], 1))
    return torch.tanh(h) if tanh else h


def norm_cancelled_keys_crn(input_nc, output_nc, noise_nc, ngf, upsample_mode="convt", n_layers_block=1, share_label_weights=True):
    keys = set()
    plan = crn_plan(input_nc, output_nc, noise_nc, ngf, upsample_mode, n_layers_block, share_label_weights)
    last = f"blockh0.1.model.{3 * (n_layers_block - 1) + 1}"
    for key, kind, cin, cout, bias in plan:
        if bias and key != last:
            keys.add(key + ".bias")
    return keys


# ----------------------------------------------------------------------------------
# the cgan training step (models/cgan_model.py:134-226)
# ----------------------------------------------------------------------------------
class CGANConfig:
    """cgan flags: unet G (`--which_model_netG unet_128|unet_256`), n_layers discriminators on cat(A, B)."""
    def __init__(self, num_downs=8, input_nc=2, output_nc=1, ngf=64, ndf=64, n_layers_D=(3, 4), scale_factor=(1, 2),
                 lambda_D=(1.0, 1.0), lambda_A=10.0, weights=None, use_dropout=True, n_layers_G_skip=-1,
                 add_gaussian_noise=False, gaussian_sigma=0.1, fineSize=512, lr=2e-4, beta1=0.5, pool_size=50,
                 no_lsgan=False, no_logD_trick=False, no_cgan=False, n_update_G=1, variant="cgan",
                 train_D_on_fake_fake_pair=False, train_G_on_fake_fake_pair=False):
        # variant "cgan2" (models/cgan2_model.py): a second label image fake_A (channels of input['B']) goes through G as
        # well, and the two flags pick which (label, generated) pair the discriminator step / the generator step uses
        self.__dict__.update(locals())
        del self.__dict__["self"]


# BASELINE configs[2] / SURVEY 8d "Config 3" (README.md:38): unet_256 ngf 64 with dropout and Gaussian noise, two scale-1
# discriminators (n_layers 3 and 4, ndf 64), weighted L1, two generator updates per discriminator update.
CGAN_README = dict(num_downs=8, ngf=64, ndf=64, n_layers_D=(3, 4), scale_factor=(1, 1), lambda_D=(0.5, 0.5), lambda_A=10.0,
                   weights=(2.0, 4.0), use_dropout=True, add_gaussian_noise=True, gaussian_sigma=0.1, fineSize=512, n_update_G=2,
                   no_lsgan=True)


class CGANOracle:
    """CGANModel restated (initialize :18-117, forward :134-139, backward_D :158-182, backward_G :184-210,
    optimize_parameters :212-226 with n_update_D = 1).  Dropout masks / Gaussian noise of the k-th generator
    forward since construction are the numpy tensors seeded 9000 + 100 k / 9500 + 100 k (make_golden.py injects the
    same ones into the reference)."""

    def __init__(self, cfg: CGANConfig, seed: int = 0):
        self.cfg = cfg
        c = cfg
        self.G = init_unet(seed + 1, c.num_downs, c.input_nc, c.output_nc, c.ngf, c.n_layers_G_skip)
        d_nc = c.output_nc if c.no_cgan else c.output_nc + c.input_nc
        self.D = [init_nlayer_d(seed + 2 + i, d_nc, c.ndf, nl, sf) for i, (nl, sf) in enumerate(zip(c.n_layers_D, c.scale_factor))]
        for v in self.G.values():
            v.requires_grad_(True)
        for d in self.D:
            for v in d.values():
                if v.is_floating_point():
                    v.requires_grad_(True)
        self.opt_G = Adam(list(self.G.values()), c.lr, c.beta1)
        self.opt_D = Adam([v for d in self.D for k, v in d.items() if k.startswith("model.")], c.lr, c.beta1)
        self.pool = ImagePool(c.pool_size)
        self.nfwd = 0

    def _g(self, a):
        c = self.cfg
        y = unet_forward(self.G, a, c.num_downs, c.ngf, c.n_layers_G_skip, c.use_dropout,
                         mask_seed=9000 + 100 * self.nfwd, add_gaussian_noise=c.add_gaussian_noise,
                         gaussian_sigma=c.gaussian_sigma, noise_seed=9500 + 100 * self.nfwd)
        self.nfwd += 1
        return y

    def forward(self):
        self.fake_B = self._g(self.real_A)
        if self.cfg.variant == "cgan2":            # cgan2_model.py:137-138: real_A first, then fake_A
            self.fake_B_from_fake_A = self._g(self.fake_A)

    def _pair(self, fake_fake):
        """(label, generated) the reference feeds a discriminator (cgan2_model.py:169-178, :200-210)."""
        if self.cfg.variant == "cgan2" and fake_fake:
            return self.fake_A, self.fake_B_from_fake_A
        return self.real_A, self.fake_B

    def _d(self, i, x):
        c = self.cfg
        return nlayer_d_forward(self.D[i], x, c.n_layers_D[i], c.scale_factor[i], use_sigmoid=c.no_lsgan)

    def backward_D(self):
        c = self.cfg
        a, b = self._pair(c.train_D_on_fake_fake_pair)
        fake = b if c.no_cgan else torch.cat((a, b), 1)
        fake = self.pool.query(fake)
        self.loss_D_fake = sum(gan_loss(self._d(i, fake.detach()), False, not c.no_lsgan) for i in range(len(self.D)))
        real = self.real_B if c.no_cgan else torch.cat((self.real_A, self.real_B), 1)
        self.loss_D_real = sum(gan_loss(self._d(i, real), True, not c.no_lsgan) for i in range(len(self.D)))
        self.loss_D = (self.loss_D_fake + self.loss_D_real) * 0.5
        self.loss_D.backward()

    def backward_G(self):
        c = self.cfg
        a, b = self._pair(c.train_G_on_fake_fake_pair)
        fake = b if c.no_cgan else torch.cat((a, b), 1)
        loss = 0
        for i, lam in enumerate(c.lambda_D):
            pred = self._d(i, fake)
            if not c.no_logD_trick:
                loss = loss + gan_loss(pred, True, not c.no_lsgan) * lam
            else:
                loss = loss - gan_loss(pred, False, not c.no_lsgan) * lam
        weight = None
        if c.weights is not None:
            weight = torch.ones(1, 1, c.fineSize, c.fineSize)
            a01 = (self.real_A.detach() + 1) / 2
            for i, wv in enumerate(c.weights):
                weight = weight + a01.narrow(1, i, 1) * (wv - 1.0)
        if c.variant == "cgan2":      # cgan2_model.py:219-232: loss_G_L1 is kept UNscaled, and there is none on the unpaired label
            self.loss_G_L1 = torch.zeros(()) if c.train_G_on_fake_fake_pair else weighted_l1(self.fake_B, self.real_B, weight)
            self.loss_G = loss + self.loss_G_L1 * c.lambda_A
        else:
            self.loss_G_L1 = weighted_l1(self.fake_B, self.real_B, weight) * c.lambda_A
            self.loss_G = loss + self.loss_G_L1
        self.loss_G.backward()

    def set_input(self, real_A, real_B, fake_A=None):
        self.real_A, self.real_B, self.fake_A = real_A, real_B, fake_A

    def _g_steps(self):
        for _ in range(self.cfg.n_update_G):
            self.opt_G.zero_grad()
            self.backward_G()
            self.opt_G.step()
            if self.cfg.n_update_G > 1:
                self.forward()          # sample_noise (:141-144): fresh dropout masks / noise

    def optimize_parameters(self):
        self.forward()
        self.opt_D.zero_grad()
        self.backward_D()
        self.opt_D.step()
        self._g_steps()

    def _gradD(self):
        return [{k: v.grad.detach().clone() for k, v in d.items() if k.startswith("model.") and v.grad is not None} for d in self.D]

    def step1_with_captures(self):
        cap = {}
        self.forward()
        cap["fake"] = self.fake_B.detach().clone()
        if self.cfg.variant == "cgan2":
            cap["fake2"] = self.fake_B_from_fake_A.detach().clone()
        self.opt_D.zero_grad()
        self.backward_D()
        cap["gradD"] = self._gradD()
        cap["loss_D"] = [float(self.loss_D_real.detach()), float(self.loss_D_fake.detach())]
        self.opt_D.step()
        self._g_steps()
        return cap

    def probe_G(self):
        """forward() + backward_G() on the initial weights."""
        self.forward()
        self.opt_G.zero_grad()
        self.opt_D.zero_grad()
        self.backward_G()
        return {"fake": self.fake_B.detach().clone(), "gradG": {k: v.grad.detach().clone() for k, v in self.G.items()},
                "gradD": self._gradD(), "loss_G": [float(self.loss_G.detach()), float(self.loss_G_L1.detach())]}

    def losses(self):
        return {"G_GAN": float(self.loss_G.detach()), "G_L1": float(self.loss_G_L1.detach()),
                "D_real": float(self.loss_D_real.detach()), "D_fake": float(self.loss_D_fake.detach())}


# ----------------------------------------------------------------------------------
# SegmentationModel (models/segm_model.py:15-263, models/loss.py:6-12): the cgan step with class logits
# ----------------------------------------------------------------------------------
class SegmConfig(CGANConfig):
    """`--model segmentation`: U-Net G emits num_classes logits (no tanh, segm_model.py:155), softmax (or `--use_sigmoid_ss`) gives
    the "fake" one-hot image the discriminators see; generator loss = GAN + (class-weighted) cross-entropy."""
    def __init__(self, use_sigmoid_ss=False, add_background_onehot=False, **kw):
        kw.setdefault("input_nc", 1)
        kw.setdefault("output_nc", 2)
        kw.setdefault("use_dropout", False)
        super().__init__(**kw)
        self.use_sigmoid_ss, self.add_background_onehot = use_sigmoid_ss, add_background_onehot
        self.label_nc = self.output_nc                                                  # channels picked from the label image
        self.output_nc = self.label_nc + 1 if add_background_onehot else self.label_nc   # num_classes (:45)


class SegmOracle(CGANOracle):
    def set_input(self, real_A, label_img):
        """segm_model.py:131-142: label image in [-1, 1] -> [0, 1] one-hot (+ background), index label = argmax."""
        b = (label_img + 1) / 2.0
        if self.cfg.add_background_onehot:
            b = torch.cat([b, 1.0 - torch.clamp(b.sum(dim=1, keepdim=True), 0, 1)], dim=1)
        self.real_A, self.real_B, self.label = real_A, b, b.max(dim=1)[1]

    def forward(self):
        c = self.cfg
        self.logit = unet_forward(self.G, self.real_A, c.num_downs, c.ngf, c.n_layers_G_skip, False, tanh=False)      # :155
        self.fake_B = torch.sigmoid(self.logit) if c.use_sigmoid_ss else F.softmax(self.logit, dim=1)               # :49,156
        self.nfwd += 1

    def backward_G(self):
        c = self.cfg
        fake = self.fake_B if c.no_cgan else torch.cat((self.real_A, self.fake_B), 1)
        self.loss_G_GAN = sum(gan_loss(self._d(i, fake), True, not c.no_lsgan) * lam for i, lam in enumerate(c.lambda_D))   # :211-213
        w = None if c.weights is None else torch.tensor(c.weights, dtype=torch.float32)
        if c.use_sigmoid_ss:                                                                                          # :216-225
            wm = None
            if w is not None:
                wm = torch.ones(1, 1, c.fineSize, c.fineSize)
                for i in range(len(c.weights)):
                    wm = wm + self.real_B.narrow(1, i, 1) * (w[i] - 1.0)
            self.loss_G_CE = F.binary_cross_entropy(self.fake_B, self.real_B, weight=wm)
        else:                                                          # CrossEntropyLoss2d = NLLLoss2d(weight)(log_softmax) (loss.py:6-12)
            self.loss_G_CE = F.nll_loss(F.log_softmax(self.logit, dim=1), self.label, weight=w)
        self.loss_G = self.loss_G_GAN + self.loss_G_CE
        self.loss_G.backward()

    def losses(self):
        return {"G_CE": float(self.loss_G_CE.detach()), "G_GAN": float(self.loss_G_GAN.detach()),
                "D_real": float(self.loss_D_real.detach()), "D_fake": float(self.loss_D_fake.detach())}


class SegmCycleConfig:
    """`--model segmentation_cycle` (models/segm_cycle_model.py): G1 image -> class logits, G2 one-hot label -> image (run on the real
    label and on G1's softmax), discriminators D2 on cat(label, image); no dropout / noise."""
    def __init__(self, input_nc=1, label_nc=2, num_downs1=7, ngf1=8, num_downs2=7, ngf2=8, ndf2=8, n_layers_D2=(3, 3), scale_factor2=(1, 2),
                 lambda_D2=(0.6, 0.4), lambda_A=1.0, lambda_B=1.0, lambda_A_cycle=1.0, weights=None, use_sigmoid_ss=False,
                 add_background_onehot=False, fineSize=256, lr1=2e-4, lr2=2e-4, beta1=0.5, pool_size=50, no_lsgan2=True, n_update_G=1):
        self.__dict__.update(locals())
        del self.__dict__["self"]
        self.num_classes = label_nc + 1 if add_background_onehot else label_nc


class SegmCycleOracle:
    """SegmentationCycleModel restated (forward :159-174, backward_D2 :201-222, backward_G :224-259, optimize_parameters :268-281
    with n_update_D2 = 1; Adam groups G1 @ lr1, G2 @ lr2, D2 @ lr2 :113-123)."""

    def __init__(self, cfg: SegmCycleConfig, seed: int = 0):
        self.cfg = c = cfg
        self.G1 = init_unet(seed + 1, c.num_downs1, c.input_nc, c.num_classes, c.ngf1, -1)
        self.G2 = init_unet(seed + 2, c.num_downs2, c.num_classes, c.input_nc, c.ngf2, -1)
        self.D = [init_nlayer_d(seed + 3 + i, c.input_nc + c.num_classes, c.ndf2, nl, sf) for i, (nl, sf) in enumerate(zip(c.n_layers_D2, c.scale_factor2))]
        for net in [self.G1, self.G2] + self.D:
            for v in net.values():
                if v.is_floating_point():
                    v.requires_grad_(True)
        self.opt_G1 = Adam(list(self.G1.values()), c.lr1, c.beta1)
        self.opt_G2 = Adam(list(self.G2.values()), c.lr2, c.beta1)
        self.opt_D = Adam([v for d in self.D for k, v in d.items() if k.startswith("model.")], c.lr2, c.beta1)
        self.pool = ImagePool(c.pool_size)

    set_input = SegmOracle.set_input

    def _g2(self, b):
        return unet_forward(self.G2, b, self.cfg.num_downs2, self.cfg.ngf2, -1, False)

    def forward(self):
        c = self.cfg
        self.logit = unet_forward(self.G1, self.real_A, c.num_downs1, c.ngf1, -1, False, tanh=False)
        self.fake_B = torch.sigmoid(self.logit) if c.use_sigmoid_ss else F.softmax(self.logit, dim=1)
        self.fake_A = self._g2(self.real_B)              # :173-174: G2 on the real label, then on G1's prediction
        self.recon_A = self._g2(self.fake_B)

    def _d(self, i, x):
        c = self.cfg
        return nlayer_d_forward(self.D[i], x, c.n_layers_D2[i], c.scale_factor2[i], use_sigmoid=c.no_lsgan2)

    def backward_D2(self):
        c = self.cfg
        fake = self.pool.query(torch.cat([self.real_B, self.fake_A], 1))
        self.loss_D2_fake = sum(gan_loss(self._d(i, fake.detach()), False, not c.no_lsgan2) for i in range(len(self.D)))
        real = torch.cat([self.real_B, self.real_A], 1)
        self.loss_D2_real = sum(gan_loss(self._d(i, real), True, not c.no_lsgan2) for i in range(len(self.D)))
        self.loss_D2 = (self.loss_D2_fake + self.loss_D2_real) * 0.5
        self.loss_D2.backward()

    def backward_G(self):
        c = self.cfg
        fake = torch.cat([self.real_B, self.fake_A], 1)
        self.loss_G2_GAN = sum(gan_loss(self._d(i, fake), True, not c.no_lsgan2) * lam for i, lam in enumerate(c.lambda_D2))
        w = None if c.weights is None else torch.tensor(c.weights, dtype=torch.float32)
        if c.use_sigmoid_ss:
            wm = None
            if w is not None:
                wm = torch.ones(1, 1, c.fineSize, c.fineSize)
                for i in range(len(c.weights)):
                    wm = wm + self.real_B.narrow(1, i, 1) * (w[i] - 1.0)
            self.loss_G1_CE = F.binary_cross_entropy(self.fake_B, self.real_B, weight=wm)
        else:
            self.loss_G1_CE = F.nll_loss(F.log_softmax(self.logit, dim=1), self.label, weight=w)
        self.loss_G_L1 = weighted_l1(self.fake_B, self.real_B, None)                  # :250
        self.loss_G_cycle = weighted_l1(self.recon_A, self.real_A, None)              # :253
        self.loss_G = self.loss_G1_CE * c.lambda_A + self.loss_G2_GAN + self.loss_G_L1 * c.lambda_B + self.loss_G_cycle * c.lambda_A_cycle
        self.loss_G.backward()

    def optimize_parameters(self):
        self.forward()
        self.opt_D.zero_grad()
        self.backward_D2()
        self.opt_D.step()
        for _ in range(self.cfg.n_update_G):
            self.opt_G1.zero_grad()
            self.opt_G2.zero_grad()
            self.backward_G()
            self.opt_G1.step()
            self.opt_G2.step()
            if self.cfg.n_update_G > 1:
                self.forward()

    def losses(self):
        return [float(v.detach()) for v in (self.loss_G1_CE, self.loss_G2_GAN, self.loss_G_L1, self.loss_G_cycle, self.loss_D2_real, self.loss_D2_fake)]


# ----------------------------------------------------------------------------------
# the twostage_cycle (DSGAN) training step (models/twostage_cycle_model.py:193-438)
# ----------------------------------------------------------------------------------
class CGANCycleConfig:
    """cgan_cycle flags with unet generators both ways (dropout off, no Gaussian noise: the step is deterministic)."""
    def __init__(self, num_downs1=7, num_downs2=7, input_nc=2, output_nc=1, ngf1=8, ngf2=8, ndf1=8, n_layers_D1=(3, 3), scale_factor1=(1, 2),
                 lambda_D1=(0.6, 0.4), lambda_A=10.0, lambda_B=10.0, lambda_A_cycle=10.0, weights=(2.0, 5.0), fineSize=256,
                 lr1=2e-4, lr2=1e-4, beta1=0.5, pool_size=50, no_lsgan1=True, no_logD_trick=False, n_update_G=1, variant="cgan_cycle",
                 train_D_on_fake_fake_pair=False, train_G_on_fake_fake_pair=False, lambda_fake_cycle=1.0):
        # variant "cgan2_cycle" (models/cgan2_cycle_model.py): second label image fake_A, five generator calls, pair-choice flags
        self.__dict__.update(locals())
        del self.__dict__["self"]


class CGANCycleOracle:
    """CGANCycleModel restated (models/cgan_cycle_model.py: forward :129-138, sample_noise :140-146, backward_D1 :162-186,
    backward_G :188-225, optimize_parameters :227-240 with n_update_D1 = 1; Adam groups G1 @ lr1, G2 @ lr2 :99-102)."""

    def __init__(self, cfg: CGANCycleConfig, seed: int = 0):
        self.cfg = c = cfg
        self.G1 = init_unet(seed + 1, c.num_downs1, c.input_nc, c.output_nc, c.ngf1, -1)
        self.G2 = init_unet(seed + 2, c.num_downs2, c.output_nc, c.input_nc, c.ngf2, -1)
        self.D = [init_nlayer_d(seed + 3 + i, c.input_nc + c.output_nc, c.ndf1, nl, sf) for i, (nl, sf) in enumerate(zip(c.n_layers_D1, c.scale_factor1))]
        for sd in [self.G1, self.G2] + self.D:
            for v in sd.values():
                if v.is_floating_point():
                    v.requires_grad_(True)
        self.opt_G1 = Adam(list(self.G1.values()), c.lr1, c.beta1)
        self.opt_G2 = Adam(list(self.G2.values()), c.lr2, c.beta1)
        self.opt_D = Adam([v for d in self.D for k, v in d.items() if k.startswith("model.")], c.lr1, c.beta1)
        self.pool = ImagePool(c.pool_size)

    def set_input(self, real_A, real_B, fake_A=None):
        self.real_A, self.real_B, self.fake_A_in = real_A, real_B, fake_A

    def _g1(self, a):
        return unet_forward(self.G1, a, self.cfg.num_downs1, self.cfg.ngf1, -1, False)

    def _g2(self, b):
        return unet_forward(self.G2, b, self.cfg.num_downs2, self.cfg.ngf2, -1, False)

    def forward(self):
        if self.cfg.variant == "cgan2_cycle":      # cgan2_cycle_model.py:123-137 (sample_noise :139-149 is the same five calls)
            self.fake_B = self._g1(self.real_A)
            self.fake_B_from_fake_A = self._g1(self.fake_A_in)
            self.fake_A = self._g2(self.real_B)                 # fake_A_from_real_B
            self.recon_A = self._g2(self.fake_B)                # recon_real_A
            self.recon_fake_A = self._g2(self.fake_B_from_fake_A)
            return
        self.fake_B = self._g1(self.real_A)
        self.fake_A = self._g2(self.real_B)
        self.recon_A = self._g2(self.fake_B)

    def sample_noise(self):
        if self.cfg.variant == "cgan2_cycle":
            return self.forward()
        self.fake_B = self._g1(self.real_A)
        self.recon_A = self._g2(self.fake_B)

    def _fake_pair(self, fake_fake):
        if self.cfg.variant == "cgan2_cycle" and fake_fake:
            return torch.cat((self.fake_A_in, self.fake_B_from_fake_A), 1)
        return torch.cat((self.real_A, self.fake_B), 1)

    def _d(self, i, x):
        c = self.cfg
        return nlayer_d_forward(self.D[i], x, c.n_layers_D1[i], c.scale_factor1[i], use_sigmoid=c.no_lsgan1)

    def backward_D1(self):
        c = self.cfg
        fake = self.pool.query(self._fake_pair(c.train_D_on_fake_fake_pair))
        self.loss_D_fake = sum(gan_loss(self._d(i, fake.detach()), False, not c.no_lsgan1) for i in range(len(self.D)))
        real = torch.cat((self.real_A, self.real_B), 1)
        self.loss_D_real = sum(gan_loss(self._d(i, real), True, not c.no_lsgan1) for i in range(len(self.D)))
        self.loss_D = (self.loss_D_fake + self.loss_D_real) * 0.5
        self.loss_D.backward()

    def backward_G(self):
        c = self.cfg
        fake = self._fake_pair(c.train_G_on_fake_fake_pair)
        gan = 0
        for i, lam in enumerate(c.lambda_D1):
            pred = self._d(i, fake)
            gan = gan + (gan_loss(pred, True, not c.no_lsgan1) * lam if not c.no_logD_trick else -gan_loss(pred, False, not c.no_lsgan1) * lam)
        weight = None
        if c.weights is not None:
            weight = torch.ones(1, 1, c.fineSize, c.fineSize)
            a01 = (self.real_A.detach() + 1) / 2
            for i, wv in enumerate(c.weights):
                weight = weight + a01.narrow(1, i, 1) * (wv - 1.0)
        self.loss_G_GAN = gan
        two = c.variant == "cgan2_cycle"
        self.loss_G_L1 = torch.zeros(()) if (two and c.train_G_on_fake_fake_pair) else weighted_l1(self.fake_B, self.real_B, weight)
        self.loss_G_CE = F.binary_cross_entropy((self.fake_A + 1) / 2, (self.real_A + 1) / 2)
        self.loss_G_cycle = F.binary_cross_entropy((self.recon_A + 1) / 2, (self.real_A + 1) / 2)
        self.loss_G = gan + self.loss_G_L1 * c.lambda_A + self.loss_G_CE * c.lambda_B + self.loss_G_cycle * c.lambda_A_cycle
        if two:     # cgan2_cycle_model.py:239-246
            self.loss_G_fake_cycle = F.binary_cross_entropy((self.recon_fake_A + 1) / 2, (self.fake_A_in + 1) / 2)
            self.loss_G = self.loss_G + self.loss_G_fake_cycle * c.lambda_A_cycle * c.lambda_fake_cycle
        self.loss_G.backward()

    def _zero_G(self):
        self.opt_G1.zero_grad()
        self.opt_G2.zero_grad()

    def _g_steps(self):
        for _ in range(self.cfg.n_update_G):
            self._zero_G()
            self.backward_G()
            self.opt_G1.step()
            self.opt_G2.step()
            if self.cfg.n_update_G > 1:
                self.sample_noise()

    def optimize_parameters(self):
        self.forward()
        self.opt_D.zero_grad()
        self.backward_D1()
        self.opt_D.step()
        self._g_steps()

    def _gradD(self):
        return [{k: v.grad.detach().clone() for k, v in d.items() if k.startswith("model.") and v.grad is not None} for d in self.D]

    def probe(self):
        """forward() + backward_D1() (before any update) + backward_G() through the initial discriminators."""
        self.forward()
        self.opt_D.zero_grad()
        self.backward_D1()
        out = {"fake_B": self.fake_B.detach().clone(), "fake_A": self.fake_A.detach().clone(), "recon_A": self.recon_A.detach().clone(),
               **({"recon_fake_A": self.recon_fake_A.detach().clone()} if self.cfg.variant == "cgan2_cycle" else {}),
               "gradD_Dstep": self._gradD(), "loss_D": [float(self.loss_D_real.detach()), float(self.loss_D_fake.detach())]}
        self.opt_D.zero_grad()
        self._zero_G()
        self.backward_G()
        out["gradG1"] = {k: v.grad.detach().clone() for k, v in self.G1.items()}
        out["gradG2"] = {k: v.grad.detach().clone() for k, v in self.G2.items()}
        out["loss_G"] = [float(self.loss_G.detach()), float(self.loss_G_GAN.detach()), float(self.loss_G_L1.detach()),
                         float(self.loss_G_CE.detach()), float(self.loss_G_cycle.detach())]
        return out

    def losses(self):
        return [float(self.loss_G.detach()), float(self.loss_G_cycle.detach()), float(self.loss_D.detach())]


class TwoStageConfig:
    """README.md:18 flags (BASELINE configs[4]) by default; binary GAN objective, no dropout."""
    def __init__(self, input_nc=2, output_nc=1, fineSize=512, ngf1=32, n_layers_G1=5, noise_nc1=8, noiseSize1=4,
                 n_layers_D1=(3, 3), ndf1=32, scale_factor1=(1, 2), lambda_D1=(0.5, 0.4), ngf2=64, upsample_mode2="bilinear",
                 n_layers_CRN_block2=2, noise_nc2=8, noiseSize2=8, nff2=32, n_layers_D2=(3, 4, 3, 4), ndf2=64,
                 scale_factor2=(1, 1, 2, 2), lambda_D2=(0.3, 0.3, 0.2, 0.2), lambda_A=10.0, lambda_B=10.0, lambda_A_cycle=5.0,
                 lambda_fake_cycle=1.0, weights=None, no_lsgan1=True, no_lsgan2=False, GAN_losses_D2=("real_fake",),
                 GAN_losses_G2=("real_fake",), lr=2e-4, lr1=2e-4, lr2=2e-4, beta1=0.5, pool_size=50, transform_1to2="bilinear_2",
                 detach_G1_from_G2_x=False, detach_G1_from_G2_y=False, no_logD_trick=False, cycle=True, lambda_G1=1.0, lambda_G2=1.0,
                 use_multi_class_GAN=False, factd=False):
        # factd (models/twostage_factD_model.py:261-292,356-383): every D2 prediction is multiplied by the x2-upsampled prediction of
        # the D1 of the same index on the (half-size) label, reflection-padded to D2's map size (util/util.py:131-145)
        # use_multi_class_GAN (twostage_cycle_model.py:86,120-146,302-335,352): D2 ends in 3 logits per pixel, classes
        # 0 = (real_A, real_B), 1 = (real_A, fake_B), 2 = (fake_A, fake_B), cross-entropy, one ImagePool per fake class
        self.__dict__.update(locals())
        del self.__dict__["self"]

    @property
    def sizeA(self):        # side of G1's label map
        return self.noiseSize1 * 2 ** (self.n_layers_G1 + 1)


class TwoStageCycleOracle:
    """TwoStageCycleModel restated (forward :193-211, backward_D1 :245-262, backward_D2_binary :264-299, backward_G :337-410,
    optimize_parameters :412-438 with one update each); G1 = fcgan, G2 = crn, F2 = unet_128 without dropout.
    cfg.cycle = False is TwoStageModel (models/twostage_model.py:185-395): no F2, plain L1, lambda_G1 / lambda_G2."""

    def __init__(self, cfg: TwoStageConfig, seed: int = 0):
        c = self.cfg = cfg
        self.G1 = init_fcgan_g(seed + 1, c.noise_nc1, c.input_nc, c.ngf1, c.n_layers_G1)
        self.G2 = init_crn(seed + 2, c.input_nc, c.output_nc, c.noise_nc2, c.ngf2, c.upsample_mode2, c.n_layers_CRN_block2, True)
        self.F2 = init_unet(seed + 3, 7, c.output_nc, c.input_nc, c.nff2, -1)
        self.D1 = [init_nlayer_d(seed + 10 + i, c.input_nc, c.ndf1, nl, sf) for i, (nl, sf) in enumerate(zip(c.n_layers_D1, c.scale_factor1))]
        self.D2 = [init_nlayer_d(seed + 20 + i, c.input_nc + c.output_nc, c.ndf2, nl, sf, 3 if c.use_multi_class_GAN else 1)
                   for i, (nl, sf) in enumerate(zip(c.n_layers_D2, c.scale_factor2))]
        for net in [self.G1, self.G2, self.F2] + self.D1 + self.D2:
            for k, v in net.items():
                if v.is_floating_point() and "running" not in k:
                    v.requires_grad_(True)
        gp = lambda net: [v for v in net.values() if v.requires_grad]
        self.opt_G = [Adam(gp(self.G1), c.lr1, c.beta1), Adam(gp(self.G2), c.lr2, c.beta1)]
        if c.cycle:
            self.opt_G.append(Adam(gp(self.F2), c.lr2, c.beta1))
        self.opt_D1 = Adam([v for d in self.D1 for k, v in d.items() if k.startswith("model.")], c.lr1, c.beta1)
        self.opt_D2 = Adam([v for d in self.D2 for k, v in d.items() if k.startswith("model.")], c.lr2, c.beta1)
        self.pool1, self.pool2 = ImagePool(c.pool_size), ImagePool(c.pool_size)
        self.pool2_1, self.pool2_2 = ImagePool(c.pool_size), ImagePool(c.pool_size)      # multi-class: one per fake class
        self.noise_iter = None      # iterator yielding (z1, z2) per forward()

    def transform(self, x):
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) if self.cfg.transform_1to2 == "bilinear_2" else x

    def transform_inverse(self, x):
        return F.avg_pool2d(x, 2, 2) if self.cfg.transform_1to2 == "bilinear_2" else x

    def _g2(self, label):
        c = self.cfg
        return crn_forward(self.G2, label, self.noise2, c.ngf2, c.upsample_mode2, c.n_layers_CRN_block2, True)

    def _f2(self, img):
        return unet_forward(self.F2, img, 7, self.cfg.nff2)

    def set_input(self, real_A, real_B):
        self.real_A, self.real_B = real_A, real_B

    def forward(self):
        c = self.cfg
        self.noise1, self.noise2 = next(self.noise_iter)
        self.fake_A = fcgan_g_forward(self.G1, self.noise1, c.n_layers_G1)
        if c.cycle:
            self.fake_A_from_real_B = self._f2(self.real_B)
        self.fake_B_from_real_A = self._g2(self.real_A)
        self.fake_B_from_fake_A = self._g2(self.transform(self.fake_A.detach() if c.detach_G1_from_G2_x else self.fake_A))
        if c.cycle:
            self.recon_real_A = self._f2(self.fake_B_from_real_A)
            self.recon_fake_A = self._f2(self.fake_B_from_fake_A)

    def _d(self, nets, cfg_nl, cfg_sf, i, x, sig):
        return nlayer_d_forward(nets[i], x, cfg_nl[i], cfg_sf[i], use_sigmoid=sig)

    def _d2(self, i, x, label=None):
        """D2_i(x); with `factd` times D1_i(label) upsampled x2 and reflection-padded (util.mul: in1 padded up to in2's size)."""
        c = self.cfg
        p2 = self._d(self.D2, c.n_layers_D2, c.scale_factor2, i, x, c.no_lsgan2)
        if not c.factd:
            return p2
        p1 = self.transform(self._d(self.D1, c.n_layers_D1, c.scale_factor1, i, label, c.no_lsgan1))
        if p1.shape == p2.shape:
            return p1 * p2
        assert p1.shape[2] < p2.shape[2] and p1.shape[3] < p2.shape[3], "util.mul returns None when the D1 map is the larger one"
        pl, pb = int((p2.shape[3] - p1.shape[3]) / 2), int((p2.shape[2] - p1.shape[2]) / 2)
        pr, pt = p2.shape[3] - p1.shape[3] - pl, p2.shape[2] - p1.shape[2] - pb
        return F.pad(p1, (pl, pr, pt, pb), mode="reflect") * p2

    def backward_D1(self):
        c = self.cfg
        fake = self.pool1.query(self.fake_A)
        real = self.transform_inverse(self.real_A)
        n = len(self.D1)
        self.loss_D1_fake = sum(gan_loss(self._d(self.D1, c.n_layers_D1, c.scale_factor1, i, fake.detach(), c.no_lsgan1), False, not c.no_lsgan1) for i in range(n))
        self.loss_D1_real = sum(gan_loss(self._d(self.D1, c.n_layers_D1, c.scale_factor1, i, real, c.no_lsgan1), True, not c.no_lsgan1) for i in range(n))
        self.loss_D1 = (self.loss_D1_fake + self.loss_D1_real) * 0.5
        self.loss_D1.backward()

    def backward_D2_multiclass(self):
        """(:302-335)"""
        c = self.cfg
        n = len(self.D2)
        d2 = lambda i, x: self._d(self.D2, c.n_layers_D2, c.scale_factor2, i, x, c.no_lsgan2)
        real = torch.cat([self.real_A, self.real_B], 1)
        self.loss_D2_0 = sum(gan_loss_multiclass(d2(i, real), 0) for i in range(n))
        fake = self.pool2_1.query(torch.cat([self.real_A, self.fake_B_from_real_A], 1))
        self.loss_D2_1 = sum(gan_loss_multiclass(d2(i, fake.detach()), 1) for i in range(n))
        fake = self.pool2_2.query(torch.cat([self.transform(self.fake_A), self.fake_B_from_fake_A], 1))
        self.loss_D2_2 = sum(gan_loss_multiclass(d2(i, fake.detach()), 2) for i in range(n))
        self.loss_D2 = (self.loss_D2_0 + self.loss_D2_1 + self.loss_D2_2) / 3
        self.loss_D2_real, self.loss_D2_fake = self.loss_D2_0, (self.loss_D2_1 + self.loss_D2_2) / 2     # for the shared reporting
        self.loss_D2.backward()

    def backward_D2(self):
        c = self.cfg
        if c.use_multi_class_GAN:
            return self.backward_D2_multiclass()
        n = len(self.D2)
        lab = lambda pair: self.transform_inverse(pair.narrow(1, 0, c.input_nc)).detach() if c.factd else None      # factD :265,276,289
        d2 = lambda i, x: self._d2(i, x, lab(x))
        self.loss_D2_fake, pairs = 0, 0
        if "real_fake" in c.GAN_losses_D2:
            fake = self.pool2.query(torch.cat([self.real_A, self.fake_B_from_real_A], 1))
            pairs += 1
            self.loss_D2_fake = self.loss_D2_fake + sum(gan_loss(d2(i, fake.detach()), False, not c.no_lsgan2) for i in range(n))
        if "fake_fake" in c.GAN_losses_D2:
            fake = self.pool2.query(torch.cat([self.transform(self.fake_A), self.fake_B_from_fake_A], 1))
            pairs += 1
            self.loss_D2_fake = self.loss_D2_fake + sum(gan_loss(d2(i, fake.detach()), False, not c.no_lsgan2) for i in range(n))
        self.loss_D2_fake = self.loss_D2_fake / pairs
        real = torch.cat([self.real_A, self.real_B], 1)
        self.loss_D2_real = sum(gan_loss(d2(i, real), True, not c.no_lsgan2) for i in range(n))
        self.loss_D2 = (self.loss_D2_fake + self.loss_D2_real) * 0.5
        self.loss_D2.backward()

    def backward_G(self):
        c = self.cfg
        bce01 = lambda x, t: F.binary_cross_entropy((x + 1) / 2, (t + 1) / 2)
        g1 = 0
        for i, lam in enumerate(c.lambda_D1):
            pred = self._d(self.D1, c.n_layers_D1, c.scale_factor1, i, self.fake_A, c.no_lsgan1)
            g1 = g1 + (gan_loss(pred, True, not c.no_lsgan1) * lam if not c.no_logD_trick else -gan_loss(pred, False, not c.no_lsgan1) * lam)
        self.loss_G1_GAN = g1
        g2, pairs = 0, 0
        fakes, labels = [], []
        if "real_fake" in c.GAN_losses_G2:
            fakes.append(torch.cat([self.real_A, self.fake_B_from_real_A], 1))
            labels.append(self.transform_inverse(self.real_A))                      # factD :359
        if "fake_fake" in c.GAN_losses_G2:
            fa = self.fake_A.detach() if c.detach_G1_from_G2_y else self.fake_A
            fakes.append(torch.cat([self.transform(fa), self.fake_B_from_fake_A], 1))
            labels.append(fa)                                                       # factD :373,376
        for fake, label in zip(fakes, labels):
            pairs += 1
            for i, lam in enumerate(c.lambda_D2):
                pred = self._d2(i, fake, label)
                if c.use_multi_class_GAN:      # flipped_label = 0 (:352); criterionGAN2(pred, False) is class 0 as well
                    g2 = g2 + (gan_loss_multiclass(pred, 0) * lam if not c.no_logD_trick else -gan_loss_multiclass(pred, 0) * lam)
                    continue
                g2 = g2 + (gan_loss(pred, True, not c.no_lsgan2) * lam if not c.no_logD_trick else -gan_loss(pred, False, not c.no_lsgan2) * lam)
        self.loss_G2_GAN = g2
        if "real_fake" in c.GAN_losses_G2:
            weight = None
            if c.weights is not None and c.cycle:
                weight = torch.ones(1, 1, c.fineSize, c.fineSize)
                a01 = (self.real_A.detach() + 1) / 2
                for i, wv in enumerate(c.weights):
                    weight = weight + a01.narrow(1, i, 1) * (wv - 1.0)
            self.loss_G2_L1 = weighted_l1(self.fake_B_from_real_A, self.real_B, weight)
        else:
            self.loss_G2_L1 = 0
        if not c.cycle:
            self.loss_G = self.loss_G1_GAN * c.lambda_G1 + self.loss_G2_GAN / pairs * c.lambda_G2 + self.loss_G2_L1 * c.lambda_G2 * c.lambda_A
            self.loss_G.backward()
            return
        self.loss_F2_CE = bce01(self.fake_A_from_real_B, self.real_A)
        self.loss_G2_real_cycle = bce01(self.recon_real_A, self.real_A)
        self.loss_G2_fake_cycle = bce01(self.recon_fake_A, self.transform(self.fake_A.detach()))
        self.loss_G = self.loss_G1_GAN + self.loss_G2_GAN / pairs + self.loss_G2_L1 * c.lambda_A + self.loss_F2_CE * c.lambda_B \
            + self.loss_G2_real_cycle * c.lambda_A_cycle + self.loss_G2_fake_cycle * c.lambda_A_cycle * c.lambda_fake_cycle
        self.loss_G.backward()

    def optimize_parameters(self):
        self.forward()
        self.opt_D1.zero_grad()
        self.backward_D1()
        self.opt_D1.step()
        self.opt_D2.zero_grad()
        self.backward_D2()
        self.opt_D2.step()
        for o in self.opt_G:
            o.zero_grad()
        self.backward_G()
        for o in self.opt_G:
            o.step()

    def _grads(self, net, prefix=""):
        return {k: v.grad.detach().clone() for k, v in net.items() if v.requires_grad and v.grad is not None and k.startswith(prefix)}

    def probe(self):
        """forward() + backward_D1() + backward_D2() + backward_G() on the initial weights, gradients captured after each
        (the G step's wasted discriminator gradients are not part of the capture)."""
        cap = {}
        self.forward()
        cap["fake_A"], cap["fake_B_from_fake_A"] = self.fake_A.detach().clone(), self.fake_B_from_fake_A.detach().clone()
        if self.cfg.cycle:
            cap["recon_fake_A"] = self.recon_fake_A.detach().clone()
        self.opt_D1.zero_grad()
        self.backward_D1()
        cap["gradD1"] = [self._grads(d, "model.") for d in self.D1]
        self.opt_D2.zero_grad()
        self.backward_D2()
        cap["gradD2"] = [self._grads(d, "model.") for d in self.D2]
        for o in self.opt_G:
            o.zero_grad()
        self.backward_G()
        cap["gradG1"], cap["gradG2"] = self._grads(self.G1), self._grads(self.G2)
        cap["gradF2"] = self._grads(self.F2) if self.cfg.cycle else {}
        cap["losses"] = self.losses()
        return cap

    def losses(self):
        f = lambda v: float(v.detach()) if torch.is_tensor(v) else float(v)
        if not self.cfg.cycle:
            return {"G2_GAN": f(self.loss_G2_GAN), "D2": f(self.loss_D2), "G1_GAN": f(self.loss_G1_GAN), "D1": f(self.loss_D1)}
        return {"G2_GAN": f(self.loss_G2_GAN), "G2_real_cycle": f(self.loss_G2_real_cycle), "G2_fake_cycle": f(self.loss_G2_fake_cycle),
                "D2": f(self.loss_D2), "G1_GAN": f(self.loss_G1_GAN), "D1": f(self.loss_D1)}


def norm_cancelled_keys_g(n_layers: int = 5):
    """Keys of FCGANGenerator whose values are numerically UNDETERMINED in the reference itself:
    conv biases that feed a BatchNorm (analytically-zero gradient => Adam turns rounding noise into
    +-lr steps) and the running_mean of that BatchNorm (tracks the bias).  Two CPU runs of the
    reference with different thread counts already disagree on these by ~lr per Adam step."""
    keys = set()
    for li in range(1, n_layers):
        keys.add(f"model.{3 * li}.bias")
        keys.add(f"model.{3 * li + 1}.running_mean")
    return keys


def norm_cancelled_keys_d(input_nc: int, ndf: int, n_layers: int):
    return {f"model.{idx}.bias" for idx, _ci, _co, _s, has_norm in nlayer_d_plan(input_nc, ndf, n_layers) if has_norm}


def grad_sample_idx(n: int, k: int = 512):
    """Indices of the elementwise-checked sample of a flattened gradient (same as make_golden.py)."""
    return np.unique(np.linspace(0, n - 1, num=min(n, k)).astype(np.int64))


def tensor_summary(t: torch.Tensor):
    t = t.detach().double()
    return [float(t.mean()), float(t.abs().max()), float(t.norm())]


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """The parity statistic of SURVEY 8d: max|a-b| / (max|b| + 1e-12)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
