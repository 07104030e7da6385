"""TestModel (`--model test`, models/test_model.py:8-45): one generator loaded from `<which_epoch>_net_G.pth`, `test()` = one
forward of the input image through it on the HIP path, visuals `real_A` / `fake_B`.  `--dataset_mode single` only
(models/models.py:31), not for training (test_model.py:13).

The reference hands `self.gpu_ids` to define_G positionally, where it lands in `n_layers_G` (test_model.py:17-20 against
networks.py:53): its generator therefore stays on the CPU whatever `--gpu_ids` says.  Here the list goes where it was meant to."""
from collections import OrderedDict

from . import networks, ops
from .base_model import BaseModel
from . import util


class TestModel(BaseModel):
    def name(self):
        return 'TestModel'

    def initialize(self, opt):
        assert not opt.isTrain
        BaseModel.initialize(self, opt)
        self.input_A = self.Tensor(opt.batchSize, opt.input_nc, opt.fineSize, opt.fineSize)
        self.netG = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, opt.which_model_netG, opt.norm, not opt.no_dropout,
                                      gpu_ids=self.gpu_ids)
        self.load_network(self.netG, 'G', opt.which_epoch)
        print('---------- Networks initialized -------------')
        networks.print_network(self.netG)
        print('-----------------------------------------------')

    def set_input(self, input):
        input_A = input['A']
        self.input_A.resize_(input_A.size()).copy_(input_A)
        self.image_paths = input['A_paths']

    def test(self):
        ops.require_gpu(self.input_A, 'TestModel.test')
        self.real_A = self.input_A
        # no autograd graph (the reference builds one and drops it); the generator stays in training mode like the reference's
        # (its Dropout layers are live at test time, test_model.py never calls eval())
        import torch
        with torch.no_grad():
            self.fake_B = self.netG.forward(self.real_A)

    def get_image_paths(self):
        return self.image_paths

    def get_current_visuals(self, save_as_single_image=True):      # the drivers pass the flag to every model (test.py:33-35)
        return OrderedDict([('real_A', util.tensor2im(self.real_A)), ('fake_B', util.tensor2im(self.fake_B))])
