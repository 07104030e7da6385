"""Option surface of the reference (options/base_options.py:11-144, options/train_options.py:4-66,
options/test_options.py) for the trainers on the MI355X path: same flag names, types, defaults and
`nargs='+'` list flags, same `parse()` protocol (gpu_ids string -> list, opt.txt dump).  Flags that only
steer out-of-scope subsystems (visdom, dataset folders) are accepted and ignored."""
import argparse
import os

import torch


class BaseOptions:
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.initialized = False
        self.isTrain = False

    def initialize(self):
        a = self.parser.add_argument
        a('--dataroot', default='synthetic', help='path to images, or "synthetic" (MI355X bench feeder)')
        a('--batchSize', type=int, default=1)
        a('--loadSize', type=int, default=286)
        a('--fineSize', type=int, default=256)
        a('--patchSize', type=int, default=70)
        a('--input_nc', type=int, default=3)
        a('--noise_nc', type=int, default=8)
        a('--noiseSize', type=int, default=1)
        a('--noiseSizeVal', type=int, default=1)
        a('--output_nc', type=int, default=3)
        a('--ngf', type=int, default=64)
        a('--ndf', type=int, default=64)
        a('--which_model_netD', type=str, default='basic')
        a('--which_model_netG', type=str, default='resnet_9blocks')
        a('--n_layers_D', type=int, default=[3], nargs='+')
        a('--n_layers_G', type=int, default=5)
        a('--scale_factor', type=int, default=[1], nargs='+')
        a('--gpu_ids', type=str, default='0')
        a('--name', type=str, default='experiment_name')
        a('--dataset_mode', type=str, default='unaligned')
        a('--model', type=str, default='cycle_gan')
        a('--which_direction', type=str, default='AtoB')
        a('--nThreads', default=2, type=int)
        a('--checkpoints_dir', type=str, default='./checkpoints')
        a('--norm', type=str, default='instance')
        a('--serial_batches', action='store_true')
        a('--display_winsize', type=int, default=256)
        a('--display_id', type=int, default=1)
        a('--display_port', type=int, default=8097)
        a('--display_single_pane_ncols', type=int, default=0)
        a('--identity', type=float, default=0.0)
        a('--no_dropout', action='store_true')
        a('--max_dataset_size', type=int, default=float("inf"))
        a('--resize_or_crop', type=str, default='resize_and_crop')
        a('--no_flip', action='store_true')
        a('--no_rotate', action='store_true')
        a('--use_residual', action='store_true')
        a('--add_gaussian_noise', action='store_true')
        a('--gaussian_sigma', type=float, default=0.1)
        a('--which_channel', type=str, default='rg')
        a('--manualSeed', type=int, default=None)
        a('--display_title', type=str, default='loss over time')
        a('--n_layers_G_skip', type=int, default=-1)
        a('--weights', type=float, default=None, nargs='+')
        a('--use_sigmoid_ss', action='store_true')            # segmentation: sigmoid instead of softmax (base_options.py:55)
        a('--which_metric', default=['None'], nargs='+')
        a('--add_background_onehot', action='store_true')
        a('--add_background_onehot_acc', action='store_true')
        a('--upsample_mode', type=str, default='convt')
        a('--no_share_label_block_weights', action='store_true')
        a('--n_layers_CRN_block', type=int, default=1)
        a('--pretrained_model_dir', type=str, default='')
        a('--transform_1to2', type=str, default='None')
        # two-stage models (options/base_options.py:62-99)
        a('--scale_factor1', type=int, default=[1], nargs='+')
        a('--scale_factor2', type=int, default=[1], nargs='+')
        a('--which_model_netD1', type=str, default='n_layers')
        a('--which_model_netG1', type=str, default='fcgan')
        a('--which_model_netF1', type=str, default='fcgan')
        a('--ngf1', type=int, default=64)
        a('--ndf1', type=int, default=64)
        a('--nff1', type=int, default=64)
        a('--n_layers_D1', type=int, default=[3], nargs='+')
        a('--n_layers_G1', type=int, default=5)
        a('--n_layers_F1', type=int, default=5)
        a('--no_dropout1', action='store_true')
        a('--noise_nc1', type=int, default=256)
        a('--noiseSize1', type=int, default=1)
        a('--which_model_netD2', type=str, default='n_layers')
        a('--which_model_netG2', type=str, default='unet_128')
        a('--which_model_netF2', type=str, default='unet_128')
        a('--ngf2', type=int, default=64)
        a('--ndf2', type=int, default=64)
        a('--nff2', type=int, default=64)
        a('--n_layers_D2', type=int, default=[3], nargs='+')
        a('--n_layers_G2', type=int, default=5)
        a('--n_layers_F2', type=int, default=5)
        a('--no_dropout2', action='store_true')
        a('--noise_nc2', type=int, default=256)
        a('--noiseSize2', type=int, default=1)
        a('--use_residual1', action='store_true')
        a('--use_residual2', action='store_true')
        a('--upsample_mode1', type=str, default='convt')
        a('--no_share_label_block_weights1', action='store_true')
        a('--n_layers_CRN_block1', type=int, default=1)
        a('--upsample_mode2', type=str, default='convt')
        a('--no_share_label_block_weights2', action='store_true')
        a('--n_layers_CRN_block2', type=int, default=1)
        a('--n_layers_G1_skip', type=int, default=-1)
        a('--n_layers_G2_skip', type=int, default=-1)
        # MI355X path extras (not in the reference)
        a('--skip_wasted_D_wgrad', action='store_true',
          help='do not compute discriminator weight gradients during the G step (the reference computes and discards them)')
        a('--hip_graph', action='store_true', help='capture the training step into hipGraphs')
        a('--no_group', action='store_true', help='launch every discriminator chain on its own instead of grouped kernels')
        a('--no_d_streams', action='store_true', help='run the discriminator chains on one stream instead of one each')
        self.initialized = True

    def parse(self, args=None, save=True, verbose=True):
        if not self.initialized:
            self.initialize()
        self.opt = self.parser.parse_args(args)
        self.opt.isTrain = self.isTrain
        str_ids = self.opt.gpu_ids.split(',')
        self.opt.gpu_ids = [int(s) for s in str_ids if int(s) >= 0]
        if len(self.opt.gpu_ids) > 0 and torch.cuda.is_available():
            torch.cuda.set_device(self.opt.gpu_ids[0])
        args_ = vars(self.opt)
        if verbose:
            print('------------ Options -------------')
            for k, v in sorted(args_.items()):
                print('%s: %s' % (str(k), str(v)))
            print('-------------- End ----------------')
        if save:
            expr_dir = os.path.join(self.opt.checkpoints_dir, self.opt.name)
            os.makedirs(expr_dir, exist_ok=True)
            with open(os.path.join(expr_dir, 'opt.txt'), 'wt') as f:
                f.write('------------ Options -------------\n')
                for k, v in sorted(args_.items()):
                    f.write('%s: %s\n' % (str(k), str(v)))
                f.write('-------------- End ----------------\n')
        return self.opt


class TrainOptions(BaseOptions):
    def initialize(self):
        BaseOptions.initialize(self)
        a = self.parser.add_argument
        a('--display_freq', type=int, default=100)
        a('--print_freq', type=int, default=100)
        a('--save_latest_freq', type=int, default=5000)
        a('--save_epoch_freq', type=int, default=5)
        a('--continue_train', action='store_true')
        a('--phase', type=str, default='train')
        a('--which_epoch', type=str, default='latest')
        a('--niter', type=int, default=100)
        a('--niter_decay', type=int, default=100)
        a('--beta1', type=float, default=0.5)
        a('--lr', type=float, default=0.0002)
        a('--no_lsgan', action='store_true')
        a('--lambda_A', type=float, default=10.0)
        a('--lambda_B', type=float, default=10.0)
        a('--n_update_G', type=int, default=1)
        a('--n_update_D', type=int, default=1)
        a('--lambda_D', type=float, default=[1.0], nargs='+')
        a('--pool_size', type=int, default=50)
        a('--no_html', action='store_true')
        a('--no_cgan', action='store_true')
        a('--noise_pool_size', type=int, default=100)
        a('--optimizer', type=str, default='adam')
        a('--pool_reject_prob', type=float, default=0.5)
        a('--train_D_on_fake_fake_pair', action='store_true')   # cgan2 (options/train_options.py:33-34)
        a('--train_G_on_fake_fake_pair', action='store_true')
        a('--no_logD_trick', action='store_true')
        a('--lambda_fake_cycle', type=float, default=1.0)
        a('--which_model_to_load', nargs='+', default=[''])
        # two-stage models (options/train_options.py:42-64)
        a('--lr1', type=float, default=0.0002)
        a('--lr2', type=float, default=0.0002)
        a('--lambda_D1', type=float, default=[1.0], nargs='+')
        a('--no_lsgan1', action='store_true')
        a('--n_update_D1', type=int, default=1)
        a('--lambda_D2', type=float, default=[1.0], nargs='+')
        a('--no_lsgan2', action='store_true')
        a('--n_update_D2', type=int, default=1)
        a('--sequential_train', action='store_true')
        a('--which_epoch_sequential', type=str, default='seq')
        a('--use_multi_class_GAN', action='store_true')
        a('--detach_G1_from_G2_x', action='store_true')
        a('--detach_G1_from_G2_y', action='store_true')
        a('--GAN_losses_D2', nargs='+', default=['real_fake'])
        a('--GAN_losses_G2', nargs='+', default=['real_fake'])
        a('--lambda_A_cycle', type=float, default=10.0)
        a('--lambda_B_cycle', type=float, default=10.0)
        a('--use_fixed_noise1', action='store_true')
        a('--lambda_G1', type=float, default=1)
        a('--lambda_G2', type=float, default=1)
        self.isTrain = True


class TestOptions(BaseOptions):
    def initialize(self):
        BaseOptions.initialize(self)
        a = self.parser.add_argument
        a('--ntest', type=int, default=float("inf"))
        a('--results_dir', type=str, default='./results/')
        a('--aspect_ratio', type=float, default=1.0)
        a('--phase', type=str, default='test')
        a('--which_epoch', type=str, default='latest')
        a('--how_many', type=int, default=50)
        a('--save_as_single_image', action='store_true')
        self.isTrain = False
