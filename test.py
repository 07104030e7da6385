#!/usr/bin/env python3
"""Sampling driver with the reference's loop (test.py:10-60): load `<which_epoch>_net_*.pth`, run `model.test()` how_many
times, write the visuals as PNGs under results_dir/name/<phase>_<which_epoch>/images/ and the index.html result page beside
them (util/visualizer.py:136-154, util/html.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from supervised_gan_amd.models import create_model  # noqa: E402
from supervised_gan_amd.options import TestOptions  # noqa: E402
from supervised_gan_amd.synthetic_data import SyntheticDataset  # noqa: E402
from supervised_gan_amd import html  # noqa: E402
from supervised_gan_amd.visualizer import Visualizer  # noqa: E402


def main(argv=None):
    opt = TestOptions().parse(argv, save=False)
    opt.nThreads, opt.batchSize, opt.serial_batches, opt.no_flip, opt.no_rotate = 1, 1, True, True, True
    model = create_model(opt)
    visualizer = Visualizer(opt)
    web_dir = os.path.join(opt.results_dir, opt.name, '%s_%s' % (opt.phase, opt.which_epoch))
    webpage = html.HTML(web_dir, 'Experiment = %s, Phase = %s, Epoch = %s' % (opt.name, opt.phase, opt.which_epoch))
    written = []

    def dump(visuals, stem):
        written.extend(visualizer.save_images(webpage, visuals, [stem + '.png']))

    if opt.model.startswith(('cgan', 'segmentation')) or opt.model == 'test':       # models that read an image (test.py:27-41; `test`: the
        # reference's loop never feeds TestModel its input, test.py:43-52 -- here it gets the dataset like the cgan models)
        if opt.dataroot == 'synthetic':
            dataset = SyntheticDataset(opt, opt.how_many)
        else:
            from supervised_gan_amd.data import create_dataset
            dataset = create_dataset(opt)
        for i, data in enumerate(dataset):
            if i >= opt.how_many:
                break
            model.set_input(data)
            model.test()
            print('process image... %s' % model.get_image_paths())
            dump(model.get_current_visuals(save_as_single_image=opt.save_as_single_image), os.path.splitext(os.path.basename(data['A_paths'][0]))[0])
    else:                                   # fcgan, twostage models (test.py:43-52)
        for i in range(opt.how_many):
            model.test()
            print('produce image... %04d.png' % (i + 1))
            dump(model.get_current_visuals(save_as_single_image=opt.save_as_single_image), '%04d' % (i + 1))
    webpage.save()
    return written


if __name__ == '__main__':
    main()
