"""BaseModel (models/base_model.py:5-64): trainer protocol + checkpoint I/O.

Checkpoint layout is the reference's: one file per network, `<epoch>_net_<label>.pth`, holding the
bare CPU state_dict with the reference's key names and logical shapes.  Loading tolerates old-torch
checkpoints (InstanceNorm running stats present, BatchNorm num_batches_tracked absent)."""
import os
from collections import OrderedDict

import torch

from . import ops


class BaseModel:
    def name(self):
        return 'BaseModel'

    def initialize(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.device = torch.device('cuda', self.gpu_ids[0]) if self.gpu_ids else torch.device('cpu')
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.model_dir = getattr(opt, 'pretrained_model_dir', '')

    def Tensor(self, *size):
        return torch.empty(*size, dtype=torch.float32, device=self.device)

    def _backward(self, loss):
        """loss.backward() with a cached unit gradient (autograd would launch a fill kernel for it on every call)."""
        one = getattr(self, '_grad_one', None)
        if one is None or one.device != loss.device or one.shape != loss.shape:
            one = self._grad_one = torch.ones_like(loss)
            ops.register_unit_grad(one)      # fused loss nodes skip the rescaling of their gradients for this tensor
        loss.backward(one)

    def set_input(self, input):
        self.input = input

    def get_current_visuals(self):
        return self.input

    def get_current_errors(self):
        return {}

    # ---- checkpoints -----------------------------------------------------------------------------
    def _checkpoint_path(self, network_label, epoch_label, model_dir):
        """`<epoch>_net_<label>.pth` under the run's directory, or under --pretrained_model_dir when the caller asks for the
        pretrained copy (any true `model_dir`: the reference ignores its value and reads opt.pretrained_model_dir, :46-49,56-59)."""
        return os.path.join(self.model_dir if model_dir else self.save_dir, '%s_net_%s.pth' % (epoch_label, network_label))

    def save_network(self, network, network_label, epoch_label, gpu_ids=[], model_dir=''):
        path = self._checkpoint_path(network_label, epoch_label, model_dir)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        # same file content as torch.save(network.cpu().state_dict(), path), without moving the live module off the GPU
        torch.save(OrderedDict((k, v.detach().to('cpu').contiguous()) for k, v in network.state_dict().items()), path)

    def load_network(self, network, network_label, epoch_label, model_dir=''):
        load_state_dict_compat(network, torch.load(self._checkpoint_path(network_label, epoch_label, model_dir), map_location='cpu'))


def _not_overridden(self, *args, **kwargs):
    return None


# the rest of the trainer protocol: hooks a subclass overrides, no-ops here (models/base_model.py:21-39,63-64)
for _hook in ('forward', 'test', 'get_image_paths', 'optimize_parameters', 'save', 'update_learning_rate'):
    setattr(BaseModel, _hook, _not_overridden)


def load_state_dict_compat(network, sd):
    """load_state_dict that accepts checkpoints written by torch <= 0.3 (the reference's authoring
    era): those carry `running_mean/var` for InstanceNorm2d(affine=False) layers (dropped here: the
    layers never used them) and lack BatchNorm `num_batches_tracked`."""
    own = network.state_dict()
    clean = OrderedDict()
    for k, v in sd.items():
        if k in own:
            clean[k] = v
        elif k.endswith('running_mean') or k.endswith('running_var'):
            continue   # stale InstanceNorm buffers
        else:
            raise KeyError('unexpected key %s in checkpoint' % k)
    missing = [k for k in own if k not in clean and not k.endswith('num_batches_tracked')]
    if missing:
        raise KeyError('missing keys in checkpoint: %s' % missing)
    network.load_state_dict(clean, strict=False)
