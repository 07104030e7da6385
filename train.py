#!/usr/bin/env python3
"""Training driver with the reference's loop (train.py:10-68): options -> create_model -> epochs of set_input /
optimize_parameters, error printing, `latest` / per-epoch checkpoints, linear LR decay after `niter` epochs.

    python train.py --dataroot synthetic --name sgan_gan --model fcgan --which_direction A --dataset_mode single --fineSize 512 \
        --input_nc 2 --which_model_netG deconv --n_layers_G 5 --ngf 32 --which_model_netD n_layers --n_layers_D 3 3 3 --ndf 32 \
        --scale_factor 1 2 4 --lambda_D 0.5 0.4 0.1 --noise_nc 8 --noiseSize 8 --norm instance --no_dropout --n_update_G 2 \
        --no_lsgan --which_channel rg            (README.md:33 with `--dataroot synthetic`)

`--dataroot synthetic` feeds device-resident random images; any other `--dataroot` is read as the reference's image folder
(`supervised_gan_amd/data.py`: single / aligned datasets, crop / flip / rotate / normalise on the device).  `--max_steps N` bounds a run."""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from supervised_gan_amd.models import create_model  # noqa: E402
from supervised_gan_amd.options import TrainOptions  # noqa: E402
from supervised_gan_amd.synthetic_data import SyntheticDataset  # noqa: E402
from supervised_gan_amd.visualizer import Visualizer  # noqa: E402


def main(argv=None):
    to = TrainOptions()
    to.initialize()
    to.parser.add_argument('--max_steps', type=int, default=0, help='stop after this many optimizer steps (0 = run all epochs)')
    to.parser.add_argument('--epoch_size', type=int, default=64, help='synthetic images per epoch')
    to.parser.add_argument('--graph', action='store_true',
                           help='replay the step as hipGraphs (graph_step.GraphedStep: ~2x fewer ms/step at bs 1 than eager launches); '
                                'the capture runs two ordinary optimizer steps on the first batch')
    opt = to.parse(argv)
    if opt.manualSeed is None:
        opt.manualSeed = random.randint(1, 10000)
    print("Random Seed: ", opt.manualSeed)
    random.seed(opt.manualSeed)
    np.random.seed(opt.manualSeed)
    torch.manual_seed(opt.manualSeed)
    if opt.dataroot == 'synthetic':
        dataset = SyntheticDataset(opt, opt.epoch_size)
    else:                                   # an image folder: <dataroot>/<phase>/*.png, transforms of data/base_dataset.py
        from supervised_gan_amd.data import create_dataset
        dataset = create_dataset(opt)
    dataset_size = len(dataset)
    print('#training images = %d' % dataset_size)
    model = create_model(opt)
    visualizer = Visualizer(opt)       # loss_log.txt + web/index.html (util/visualizer.py)
    graphed = None
    if opt.graph:
        from supervised_gan_amd.graph_step import GraphedStep
        graphed = GraphedStep(model)
    total_steps = 0
    for epoch in range(1, opt.niter + opt.niter_decay + 1):
        epoch_start_time = time.time()
        for data in dataset:
            iter_start_time = time.time()
            total_steps += opt.batchSize
            epoch_iter = total_steps - dataset_size * (epoch - 1)
            if graphed is None:
                model.set_input(data)
                model.optimize_parameters()
            elif not graphed._captured:
                graphed.capture(data)          # = warmup_steps ordinary steps on this batch
            else:
                graphed.step(data)
            if total_steps % opt.display_freq == 0:
                visualizer.display_current_results(model.get_current_visuals(), epoch)
            if total_steps % opt.print_freq == 0:
                visualizer.print_current_errors(epoch, epoch_iter, model.get_current_errors(), (time.time() - iter_start_time) / opt.batchSize)
            if total_steps % opt.save_latest_freq == 0:
                print('saving the latest model (epoch %d, total_steps %d)' % (epoch, total_steps))
                model.save('latest')
            if opt.max_steps and total_steps >= opt.max_steps:
                model.save('latest')
                return model
        if epoch % opt.save_epoch_freq == 0:
            print('saving the model at the end of epoch %d, iters %d' % (epoch, total_steps))
            model.save('latest')
            model.save(epoch)
        print('End of epoch %d / %d \t Time Taken: %d sec' % (epoch, opt.niter + opt.niter_decay, time.time() - epoch_start_time))
        if epoch > opt.niter:
            model.update_learning_rate()
    return model


if __name__ == '__main__':
    main()
