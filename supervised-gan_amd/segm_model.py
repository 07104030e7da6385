"""SegmentationModel (models/segm_model.py:15-341): the conditional-GAN step with the generator emitting class logits -- image
real_A -> U-Net -> logits -> softmax (or `--use_sigmoid_ss` sigmoid) = the "fake" one-hot label the PatchGAN discriminators see on
cat(real_A, .); generator loss = sum_i lambda_i GAN(D_i(fake), 1) + (class-weighted) cross-entropy against the label.

Built on CGANModel: discriminator step, pooling, optimizers, checkpoints and the hipGraph step are inherited; the generator runs
with the caller's `activation=` (identity) so the conv chain ends raw, and softmax / cross-entropy are PyTorch's own kernels on
the num_classes x H x W maps (models/loss.py:6-12 is NLLLoss2d(log_softmax))."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import networks
from .cgan_model import CGANModel
from .losses import cross_entropy_logits, softmax_channels
from .util import compute_Rand_F_scores


def _identity(x):
    return x


class SegmentationModel(CGANModel):
    def name(self):
        return 'SegmentationModel'

    def initialize(self, opt):
        idx = {'r': 0, 'g': 1, 'b': 2}
        picks = [[idx[c] for c in part] for part in opt.which_channel.split('_')]
        assert len(picks) == 2
        self.label_nc = len(picks[1])
        self.num_classes = self.label_nc + 1 if opt.add_background_onehot else self.label_nc          # segm_model.py:45
        if getattr(opt, 'which_model_netD', 'None') == 'None' and opt.isTrain:
            raise NotImplementedError("--which_model_netD None (cross-entropy only) is not on the MI355X path")
        CGANModel.initialize(self, opt)
        self.class_weights = None if opt.weights is None else torch.tensor(opt.weights, dtype=torch.float32, device=self.device)
        self.use_sigmoid_ss = opt.use_sigmoid_ss
        self.reset_accs()

    def _output_channels(self, opt):
        return self.num_classes          # generator output and discriminator input are sized by the class count (:69,83-86)

    # ---- data ---------------------------------------------------------------------------------
    def set_input(self, input):
        """segm_model.py:120-143: label channels rescaled to [0, 1], optional background class, index label = argmax."""
        CGANModel.set_input(self, input)
        self._one_hot_label()

    def _one_hot_label(self):
        b = (self.input_B[:, :self.label_nc] + 1) / 2.0
        if self.opt.add_background_onehot:
            b = torch.cat([b, 1.0 - torch.clamp(b.sum(dim=1, keepdim=True), 0, 1)], dim=1)
        self.input_B.resize_(b.size()).copy_(b)
        lab = b.max(dim=1)[1]
        if getattr(self, 'label', None) is None or self.label.shape != lab.shape:
            self.label = torch.empty_like(lab)        # persistent: a captured hipGraph keeps reading this buffer (label_, :61,138-139)
        self.label.copy_(lab)

    def forward(self):
        self.real_A = self.input_A
        self.real_B = self.input_B
        self.noise = self._draw_noise()
        self.logit = self.netG.forward(self.real_A, self.noise, activation=_identity)                 # :155
        self.fake_B = torch.sigmoid(self.logit) if self.use_sigmoid_ss else softmax_channels(self.logit)

    sample_noise = forward

    def test(self):
        with torch.no_grad():
            self.forward()

    # ---- losses -------------------------------------------------------------------------------
    def compute_cross_entropy_loss(self, weighted=False):
        if self.use_sigmoid_ss:                                                                           # :216-225, :235-236
            wm = None
            if weighted and self.class_weights is not None:
                wm = torch.ones_like(self.real_B[:, :1])
                for i in range(self.class_weights.numel()):
                    wm = wm + self.real_B.narrow(1, i, 1) * (self.class_weights[i] - 1.0)
            self.loss_G_CE = F.binary_cross_entropy(self.fake_B, self.real_B, weight=wm)
        else:
            w = self.class_weights if (weighted or self.isTrain) else None
            self.loss_G_CE = cross_entropy_logits(self.logit, self.label, 0, w)      # models/loss.py:6-12 on the HIP kernel
        return self.loss_G_CE

    def backward_G(self):
        """loss_G = sum_i lambda_i * GAN(D_i(cat(A, fake_B)), 1) + CE   (segm_model.py:203-232)"""
        skip = getattr(self.opt, 'skip_wasted_D_wgrad', False)
        for netD in self.netD:
            netD.compute_param_grads = not skip
        fake = self.fake_B if self.opt.no_cgan else networks.cat_pair(self.real_A, self.fake_B)
        self.loss_G_GAN, self._each_G = self._d_losses([(d, fake, True) for d in self.netD], list(self.opt.lambda_D))
        for netD in self.netD:
            netD.compute_param_grads = True
        self.loss_G = self.loss_G_GAN + self.compute_cross_entropy_loss(weighted=True)
        self._backward(self.loss_G)

    def get_current_errors(self):
        return OrderedDict([('G_CE', float(self.loss_G_CE.detach())), ('G_GAN', float(self.loss_G_GAN.detach())),
                            ('D_real', float(self.loss_D_real)), ('D_fake', float(self.loss_D_fake))])

    def get_current_visuals(self, save_as_single_image=False):
        three = lambda t: t if t.shape[1] in (1, 3) else torch.cat([t, torch.zeros_like(t[:, :1])], 1)[:, :3]      # noqa: E731
        return OrderedDict([('image', self.real_A.detach()), ('label', three(self.real_B.detach() * 2 - 1)),
                            ('prediction', three(self.fake_B.detach() * 2 - 1))])

    # ---- accuracy (segm_model.py:265-341) ---------------------------------------------------------------------------
    def reset_accs(self):
        self.confusion, self.numAveragedPixels, self.numAveragedImages = 0, 0, 0
        self.pixelAcc = self.meanAcc = self.meanIU = self.RandScore = 0

    def accum_accs(self):
        if 'RandScore' in self.opt.which_metric:
            self.compute_current_Rand_score()
        if 'meanIU' in self.opt.which_metric:
            self.compute_current_accuracy()

    def compute_current_Rand_score(self):
        """Running mean of the Rand F-score of the predicted against the true boundary map (segm_model.py:299-307): a host metric
        (connected components of a 512x512 map), computed on the CPU copy like the reference does."""
        assert self.num_classes == 2      # binary segmentation only, as in the reference
        RIs = compute_Rand_F_scores(self.fake_B.detach().cpu().numpy(), self.real_B.detach().cpu().numpy(), do_thin=False)
        n = self.numAveragedImages
        self.numAveragedImages = n + RIs.size
        self.RandScore = (n * self.RandScore + RIs.sum()) / self.numAveragedImages

    def compute_current_accuracy(self):
        if self.opt.add_background_onehot_acc:
            bg = lambda t: torch.cat([t, 1.0 - torch.clamp(t.sum(dim=1, keepdim=True), max=1)], 1).argmax(dim=1)      # noqa: E731
            labels, pred, k = bg(self.real_B), bg(self.fake_B.detach()), self.num_classes + 1
        else:
            labels, pred, k = self.label, self.logit.detach().argmax(dim=1), self.num_classes
        conf = torch.bincount((labels.reshape(-1) * k + pred.reshape(-1)), minlength=k * k).reshape(k, k).double().cpu().numpy()
        self.confusion = self.confusion + conf
        self.numAveragedPixels += labels.numel()
        rel, sel, tp = self.confusion.sum(axis=1), self.confusion.sum(axis=0), np.diag(self.confusion)
        self.pixelAcc = tp.sum() / max(1, self.numAveragedPixels)
        self.meanAcc = float(np.mean(tp / np.maximum(1, rel)))
        self.meanIU = float(np.mean(tp / np.maximum(1, rel + sel - tp)))

    def get_current_accs(self):
        return OrderedDict(([('RandScore', self.RandScore)] if 'RandScore' in self.opt.which_metric else [])
                           + ([('meanIU', self.meanIU)] if 'meanIU' in self.opt.which_metric else []))
