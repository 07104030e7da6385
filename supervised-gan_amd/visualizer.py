"""Visualizer of the drivers (util/visualizer.py:9-154): `loss_log.txt`, the per-epoch image dump + `web/index.html` of a
training run (display_current_results), and the result page of test.py (save_images).  The visdom panes of the reference
(`--display_id > 0`) need a visdom server and are not part of this path: `display_id` is accepted and ignored."""
import ntpath
import os
import time

from . import html
from .util import save_image, tensor2im


def _as_image(v):
    """The trainers hand back tensors ([1, C, H, W] in [-1, 1]); the reference's get_current_visuals already holds uint8 arrays."""
    return tensor2im(v) if hasattr(v, 'detach') else v


class Visualizer:
    def __init__(self, opt):
        self.use_html = opt.isTrain and not opt.no_html
        self.win_size = opt.display_winsize
        self.name = opt.name
        if self.use_html:
            self.web_dir = os.path.join(opt.checkpoints_dir, opt.name, 'web')
            self.img_dir = os.path.join(self.web_dir, 'images')
            print('create web directory %s...' % self.web_dir)
            os.makedirs(self.img_dir, exist_ok=True)
        self.log_name = os.path.join(opt.checkpoints_dir, opt.name, 'loss_log.txt')
        os.makedirs(os.path.dirname(self.log_name), exist_ok=True)
        with open(self.log_name, "a") as log_file:
            log_file.write('================ Training Loss (%s) ================\n' % time.strftime("%c"))

    def display_current_results(self, visuals, epoch):
        """Save this epoch's visuals as images/epoch%.3d_<label>.png and rebuild index.html with every epoch so far, newest first."""
        if not self.use_html:
            return
        for label, v in visuals.items():
            save_image(_as_image(v), os.path.join(self.img_dir, 'epoch%.3d_%s.png' % (epoch, label)))
        webpage = html.HTML(self.web_dir, 'Experiment name = %s' % self.name, reflesh=1)
        for n in range(epoch, 0, -1):
            webpage.add_header('epoch [%d]' % n)
            ims = ['epoch%.3d_%s.png' % (n, label) for label in visuals]
            webpage.add_images(ims, list(visuals), ims, width=self.win_size)
        webpage.save()

    def print_current_errors(self, epoch, i, errors, t):
        message = '(epoch: %d, iters: %d, time: %.3f) ' % (epoch, i, t) + ''.join('%s: %.3f ' % kv for kv in errors.items())
        print(message)
        with open(self.log_name, "a") as log_file:
            log_file.write('%s\n' % message)

    def save_images(self, webpage, visuals, image_path):
        """One header + one row of `<name>_<label>.png` on `webpage` (test.py's result page); returns the written paths."""
        image_dir = webpage.get_image_dir()
        name = os.path.splitext(ntpath.basename(image_path[0]))[0]
        webpage.add_header(name)
        ims, written = [], []
        for label, v in visuals.items():
            image_name = '%s_%s.png' % (name, label)
            save_image(_as_image(v), os.path.join(image_dir, image_name))
            ims.append(image_name)
            written.append(os.path.join(image_dir, image_name))
        webpage.add_images(ims, list(visuals), ims, width=self.win_size)
        return written
