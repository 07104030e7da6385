"""train.py / test.py (the reference's driver loops, train.py:10-68, test.py:10-60) on the MI355X path with the synthetic feeder:
a few optimizer steps, checkpoints in the reference's layout, then sampling from those checkpoints into PNG files."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")


def test_train_then_test_fcgan(tmp_path):
    _need_gpu()
    import test as test_driver
    import train as train_driver
    net = ["--name", "drv_fcgan", "--model", "fcgan", "--which_direction", "A", "--dataset_mode", "single", "--fineSize", "128",
           "--input_nc", "2", "--which_model_netG", "deconv", "--n_layers_G", "5", "--ngf", "8", "--noise_nc", "8", "--noiseSize", "2",
           "--norm", "instance", "--no_dropout", "--which_channel", "rg", "--gpu_ids", "0", "--checkpoints_dir", str(tmp_path / "ckpt"),
           "--dataroot", "synthetic", "--manualSeed", "3"]
    m = train_driver.main(net + ["--which_model_netD", "n_layers", "--n_layers_D", "3", "3", "--ndf", "8", "--scale_factor", "1", "2",
                                 "--lambda_D", "0.6", "0.4", "--n_update_G", "2", "--no_lsgan", "--max_steps", "3", "--print_freq", "1"])
    assert m.optimizer_D.step_count == 3 and m.optimizer_G.step_count == 6
    files = sorted(os.listdir(tmp_path / "ckpt" / "drv_fcgan"))
    assert [f for f in files if f.endswith(".pth")] == ["latest_net_D_0.pth", "latest_net_D_1.pth", "latest_net_G.pth"]
    log = open(tmp_path / "ckpt" / "drv_fcgan" / "loss_log.txt").read().splitlines()          # util/visualizer.py:126-133
    assert log[0].startswith("================ Training Loss") and len(log) == 4 and log[3].startswith("(epoch: 1, iters: 3, time: ")
    assert "G_GAN: " in log[3] and "D_real: " in log[3] and "D_fake: " in log[3]
    out = test_driver.main(net + ["--results_dir", str(tmp_path / "res"), "--how_many", "2"])
    assert len(out) == 2 and all(os.path.exists(p) for p in out)
    from PIL import Image
    im = np.asarray(Image.open(out[0]))
    assert im.shape == (128, 128, 3) and im[..., 2].max() == 0          # 2-channel label image: blue plane is zero


def test_train_then_test_fcgan_star(tmp_path):
    """`--which_model_netG fcgan_star` (models/networks.py:89-91) through the fcgan trainer, eager and as hipGraphs, then test.py."""
    _need_gpu()
    import test as test_driver
    import train as train_driver
    net = ["--name", "drv_star", "--model", "fcgan", "--which_direction", "A", "--dataset_mode", "single", "--fineSize", "128",
           "--input_nc", "2", "--which_model_netG", "fcgan_star", "--n_layers_G", "5", "--ngf", "8", "--noise_nc", "8", "--noiseSize", "2",
           "--norm", "instance", "--no_dropout", "--which_channel", "rg", "--gpu_ids", "0", "--checkpoints_dir", str(tmp_path / "ckpt"),
           "--dataroot", "synthetic", "--manualSeed", "3"]
    d = ["--which_model_netD", "n_layers", "--n_layers_D", "3", "3", "--ndf", "8", "--scale_factor", "1", "2", "--lambda_D", "0.6", "0.4",
         "--n_update_G", "2", "--no_lsgan", "--max_steps", "3", "--print_freq", "1"]
    for extra in ([], ["--graph"]):
        m = train_driver.main(net + d + extra)
        torch.cuda.synchronize()
        n = 4 if extra else 3            # --graph: two capture steps on the first batch, then replays
        assert m.optimizer_D.step_count == n and m.optimizer_G.step_count == 2 * n
        assert all(np.isfinite(v) for v in m.get_current_errors().values())
        assert int(m.netG.state_dict()["conv3b.1.num_batches_tracked"]) >= 6
    out = test_driver.main(net + ["--results_dir", str(tmp_path / "res"), "--how_many", "2"])
    assert len(out) == 2 and all(os.path.exists(p) for p in out)


@pytest.mark.parametrize("netG", ["unet_128", "resnet_9blocks"])
def test_train_then_test_cgan(tmp_path, netG):
    _need_gpu()
    import test as test_driver
    import train as train_driver
    net = ["--name", "drv_cgan", "--model", "cgan", "--which_direction", "AtoB", "--dataset_mode", "single", "--fineSize", "256",
           "--which_model_netG", netG, "--ngf", "8", "--norm", "instance", "--which_channel", "rg_b", "--gpu_ids", "0",
           "--checkpoints_dir", str(tmp_path / "ckpt"), "--dataroot", "synthetic", "--manualSeed", "4"]
    m = train_driver.main(net + ["--which_model_netD", "n_layers", "--n_layers_D", "3", "4", "--ndf", "8", "--scale_factor", "1", "1",
                                 "--lambda_D", "0.5", "0.5", "--weights", "2", "4", "--no_lsgan", "--max_steps", "2", "--print_freq", "1",
                                 "--display_freq", "1"])
    assert all(np.isfinite(v) for v in m.get_current_errors().values())
    web = tmp_path / "ckpt" / "drv_cgan" / "web"          # the training run's result page (util/visualizer.py:77-93)
    assert (web / "index.html").exists() and any(f.startswith("epoch001_") for f in os.listdir(web / "images"))
    out = test_driver.main(net + ["--results_dir", str(tmp_path / "res"), "--how_many", "2"])
    assert len(out) == 4 and all(os.path.exists(p) for p in out)            # real_A + fake_B per image
    assert (tmp_path / "res" / "drv_cgan" / "test_latest" / "index.html").exists()
    if netG == "unet_128":      # --model test (models/test_model.py): the saved generator alone, fed by the single-image dataset
        args = [a if a != "cgan" else "test" for a in net] + ["--input_nc", "2", "--output_nc", "1"]
        out2 = test_driver.main(args + ["--results_dir", str(tmp_path / "res2"), "--how_many", "2", "--no_dropout"])
        assert len(out2) == 4 and all(os.path.exists(p) for p in out2)
        from supervised_gan_amd.models import create_model
        from supervised_gan_amd.options import TestOptions
        opt = TestOptions().parse(args + ["--no_dropout"], save=False, verbose=False)
        tm = create_model(opt)
        assert tm.name() == "TestModel"
        x = torch.rand(1, 2, 256, 256) * 2 - 1
        tm.set_input({"A": x, "A_paths": ["x.png"]})
        tm.test()
        y1 = tm.fake_B.clone()
        tm.test()
        torch.cuda.synchronize()
        assert y1.shape == (1, 1, 256, 256) and torch.equal(y1, tm.fake_B) and float(y1.abs().max()) <= 1.0      # no dropout: repeatable
        sd = torch.load(tmp_path / "ckpt" / "drv_cgan" / "latest_net_G.pth", map_location="cpu")
        for k, v in tm.netG.state_dict().items():      # it IS the trained generator
            assert torch.equal(v.cpu(), sd[k]), k
        assert tm.get_image_paths() == ["x.png"] and set(tm.get_current_visuals()) == {"real_A", "fake_B"}


def test_train_cgan2(tmp_path):
    """`--model cgan2` (two label images per sample, unaligned feeder) through train.py, with either pair choice."""
    _need_gpu()
    import train as train_driver
    net = ["--name", "drv_cgan2", "--model", "cgan2", "--dataset_mode", "unaligned", "--fineSize", "256",
           "--which_model_netG", "unet_128", "--ngf", "8", "--norm", "instance", "--which_channel", "rg_b", "--gpu_ids", "0",
           "--checkpoints_dir", str(tmp_path / "ckpt"), "--dataroot", "synthetic", "--manualSeed", "4",
           "--which_model_netD", "n_layers", "--n_layers_D", "3", "3", "--ndf", "8", "--scale_factor", "1", "2",
           "--lambda_D", "0.5", "0.5", "--weights", "2", "4", "--no_lsgan", "--max_steps", "3", "--print_freq", "1"]
    for extra in ([], ["--train_D_on_fake_fake_pair", "--train_G_on_fake_fake_pair"]):
        m = train_driver.main(net + extra)
        assert all(np.isfinite(v) for v in m.get_current_errors().values())
        assert m.fake_B_from_fake_A.shape == (1, 1, 256, 256)


@pytest.mark.parametrize("model,mode", [("cgan_cycle", "single"), ("cgan2_cycle", "unaligned")])
def test_train_then_test_cgan_cycle(tmp_path, model, mode):
    """`--model cgan_cycle` / `cgan2_cycle` through train.py (checkpoints G1, G2, D1_n) and test.py (sampling from them)."""
    _need_gpu()
    import test as test_driver
    import train as train_driver
    name = "drv_" + model
    net = ["--name", name, "--model", model, "--which_direction", "AtoB", "--dataset_mode", mode, "--fineSize", "256",
           "--which_model_netG1", "unet_128", "--ngf1", "8", "--which_model_netG2", "unet_128", "--ngf2", "8", "--norm", "instance",
           "--which_channel", "rg_b", "--gpu_ids", "0", "--checkpoints_dir", str(tmp_path / "ckpt"), "--dataroot", "synthetic",
           "--manualSeed", "4", "--no_dropout1", "--no_dropout2"]
    m = train_driver.main(net + ["--which_model_netD1", "n_layers", "--n_layers_D1", "3", "3", "--ndf1", "8", "--scale_factor1", "1", "2",
                                 "--lambda_D1", "0.6", "0.4", "--weights", "2", "4", "--no_lsgan1", "--max_steps", "2", "--print_freq", "1"])
    assert all(np.isfinite(v) for v in m.get_current_errors().values())
    files = sorted(f for f in os.listdir(tmp_path / "ckpt" / name) if f.endswith(".pth"))
    assert files == ["latest_net_D1_0.pth", "latest_net_D1_1.pth", "latest_net_G1.pth", "latest_net_G2.pth"]
    out = test_driver.main(net + ["--results_dir", str(tmp_path / "res"), "--how_many", "2"])
    assert len(out) == 4 and all(os.path.exists(p) for p in out)            # real_A + fake_B per image


def test_train_graphed(tmp_path):
    """train.py --graph: the driver loop on hipGraph replays (capture on the first batch, then replays)."""
    _need_gpu()
    import train as train_driver
    net = ["--name", "drv_graph", "--model", "fcgan", "--which_direction", "A", "--dataset_mode", "single", "--fineSize", "128",
           "--input_nc", "2", "--which_model_netG", "deconv", "--n_layers_G", "5", "--ngf", "8", "--noise_nc", "8", "--noiseSize", "2",
           "--norm", "instance", "--no_dropout", "--which_channel", "rg", "--gpu_ids", "0", "--checkpoints_dir", str(tmp_path / "ckpt"),
           "--dataroot", "synthetic", "--manualSeed", "3", "--which_model_netD", "n_layers", "--n_layers_D", "3", "3", "--ndf", "8",
           "--scale_factor", "1", "2", "--lambda_D", "0.6", "0.4", "--n_update_G", "2", "--no_lsgan", "--max_steps", "6", "--print_freq", "1",
           "--graph"]
    m = train_driver.main(net)
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in m.get_current_errors().values())
    assert m.optimizer_D.step_count == 2 + 5            # two capture steps on the first batch, then five replays
    assert os.path.exists(tmp_path / "ckpt" / "drv_graph" / "latest_net_G.pth")


def test_train_then_test_segmentation(tmp_path):
    """`--model segmentation` (models/segm_model.py) through train.py -- eager and as hipGraphs -- and test.py."""
    _need_gpu()
    import test as test_driver
    import train as train_driver
    net = ["--name", "drv_segm", "--model", "segmentation", "--which_direction", "AtoB", "--dataset_mode", "aligned", "--fineSize", "256",
           "--which_model_netG", "unet_128", "--ngf", "8", "--norm", "instance", "--which_channel", "b_rg", "--gpu_ids", "0", "--no_dropout",
           "--checkpoints_dir", str(tmp_path / "ckpt"), "--dataroot", "synthetic", "--manualSeed", "4"]
    d = ["--which_model_netD", "n_layers", "--n_layers_D", "3", "--ndf", "8", "--scale_factor", "1", "--lambda_D", "1.0", "--weights", "1", "2",
         "--no_lsgan", "--max_steps", "3", "--print_freq", "1"]
    for extra in ([], ["--graph"]):
        m = train_driver.main(net + d + extra)
        torch.cuda.synchronize()
        e = m.get_current_errors()
        assert list(e) == ["G_CE", "G_GAN", "D_real", "D_fake"] and all(np.isfinite(v) for v in e.values())
    out = test_driver.main(net + ["--results_dir", str(tmp_path / "res"), "--how_many", "2"])
    assert len(out) == 6 and all(os.path.exists(p) for p in out)            # image, label, prediction per sample


def test_train_then_test_segmentation_cycle(tmp_path):
    """`--model segmentation_cycle` (models/segm_cycle_model.py) through train.py -- eager and as hipGraphs -- and test.py."""
    _need_gpu()
    import test as test_driver
    import train as train_driver
    net = ["--name", "drv_segc", "--model", "segmentation_cycle", "--which_direction", "AtoB", "--dataset_mode", "aligned", "--fineSize", "256",
           "--which_model_netG1", "unet_128", "--ngf1", "8", "--which_model_netG2", "unet_128", "--ngf2", "8", "--norm", "instance",
           "--which_channel", "b_rg", "--gpu_ids", "0", "--no_dropout1", "--no_dropout2", "--checkpoints_dir", str(tmp_path / "ckpt"),
           "--dataroot", "synthetic", "--manualSeed", "4"]
    d = ["--which_model_netD2", "n_layers", "--n_layers_D2", "3", "--ndf2", "8", "--scale_factor2", "1", "--lambda_D2", "1.0", "--no_lsgan2",
         "--max_steps", "3", "--print_freq", "1"]
    for extra in ([], ["--graph"]):
        m = train_driver.main(net + d + extra)
        torch.cuda.synchronize()
        e = m.get_current_errors()
        assert list(e) == ["G_CE", "G_GAN", "G_L1", "G_cycle", "D_real", "D_fake"] and all(np.isfinite(v) for v in e.values())
    files = sorted(f for f in os.listdir(tmp_path / "ckpt" / "drv_segc") if f.endswith(".pth"))
    assert files == ["latest_net_D2_0.pth", "latest_net_G1.pth", "latest_net_G2.pth"]
    out = test_driver.main(net + ["--results_dir", str(tmp_path / "res"), "--how_many", "2"])
    assert len(out) == 6 and all(os.path.exists(p) for p in out)


@pytest.mark.parametrize("model", ["twostage", "twostage_factd"])
def test_train_then_test_twostage(tmp_path, model):
    """`--model twostage` / `twostage_factd` (models/twostage_model.py, twostage_factD_model.py) through train.py and test.py; the
    factored variant needs D1 maps that fit inside D2's after the x2 upsampling (n_layers_D1 4 against n_layers_D2 3)."""
    _need_gpu()
    import test as test_driver
    import train as train_driver
    net = ["--name", "drv_" + model, "--model", model, "--which_direction", "AtoB", "--dataset_mode", "aligned", "--fineSize", "256",
           "--which_channel", "rg_b", "--which_model_netG1", "fcgan", "--n_layers_G1", "5", "--ngf1", "8", "--noise_nc1", "8", "--noiseSize1", "2",
           "--which_model_netG2", "crn", "--ngf2", "8", "--noise_nc2", "8", "--noiseSize2", "4", "--upsample_mode2", "bilinear",
           "--n_layers_CRN_block2", "2", "--transform_1to2", "bilinear_2", "--norm", "instance", "--no_dropout1", "--no_dropout2", "--gpu_ids", "0",
           "--checkpoints_dir", str(tmp_path / "ckpt"), "--dataroot", "synthetic", "--manualSeed", "5"]
    d = ["--which_model_netD1", "n_layers", "--n_layers_D1", "4", "4", "--ndf1", "8", "--scale_factor1", "1", "2", "--lambda_D1", "0.5", "0.4",
         "--which_model_netD2", "n_layers", "--n_layers_D2", "3", "3", "--ndf2", "8", "--scale_factor2", "1", "2", "--lambda_D2", "0.6", "0.4",
         "--no_lsgan1", "--no_lsgan2", "--GAN_losses_D2", "real_fake", "fake_fake", "--GAN_losses_G2", "real_fake", "fake_fake",
         "--max_steps", "3", "--print_freq", "1"]
    m = train_driver.main(net + d)
    torch.cuda.synchronize()
    assert m.name() == {"twostage": "TwoStageModel", "twostage_factd": "TwoStageFactDModel"}[model]
    assert all(np.isfinite(v) for v in m.get_current_errors().values())
    if model == "twostage_factd":      # the factored terms (discriminator calls + upsample / pad / product / loss on torch's kernels) capture as well
        mg = train_driver.main(net + d + ["--graph"])
        torch.cuda.synchronize()
        assert all(np.isfinite(v) for v in mg.get_current_errors().values())
    out = test_driver.main(net + ["--results_dir", str(tmp_path / "res"), "--how_many", "2"])
    assert out and all(os.path.exists(p) for p in out)


def _write_images(folder, n, w, h, seed):
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    rng = np.random.RandomState(seed)
    for i in range(n):
        Image.fromarray(rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8), "RGB").save(os.path.join(folder, "img_%02d.png" % i))


def test_train_and_test_from_image_folders(tmp_path):
    """`--dataroot <folder>`: the reference's single and aligned datasets (data/single_dataset.py, data/aligned_dataset.py) with their
    transforms on the device; fcgan trains from <root>/train/*.png, cgan from side-by-side A|B files and samples from <root>/test."""
    _need_gpu()
    import test as test_driver
    import train as train_driver
    from supervised_gan_amd.data import SingleFolderDataset
    from supervised_gan_amd.options import TrainOptions
    root = tmp_path / "data"
    _write_images(str(root / "single" / "train"), 5, 150, 140, 1)
    common = ["--gpu_ids", "0", "--checkpoints_dir", str(tmp_path / "ckpt"), "--manualSeed", "3", "--norm", "instance", "--print_freq", "1"]
    net = ["--name", "fold_fcgan", "--model", "fcgan", "--which_direction", "A", "--dataset_mode", "single", "--fineSize", "128",
           "--loadSize", "143", "--input_nc", "2", "--which_model_netG", "deconv", "--n_layers_G", "5", "--ngf", "8", "--noise_nc", "8",
           "--noiseSize", "2", "--no_dropout", "--which_channel", "rg", "--dataroot", str(root / "single"),
           "--which_model_netD", "n_layers", "--n_layers_D", "3", "3", "--ndf", "8", "--scale_factor", "1", "2", "--lambda_D", "0.6", "0.4",
           "--no_lsgan"]
    m = train_driver.main(net + common + ["--max_steps", "4"])
    assert m.optimizer_D.step_count == 4 and all(np.isfinite(v) for v in m.get_current_errors().values())
    opt = TrainOptions().parse(net + common, save=False)
    item = next(iter(SingleFolderDataset(opt)))
    assert tuple(item["A"].shape) == (1, 3, 128, 128) and item["A"].is_cuda and float(item["A"].abs().max()) <= 1.0
    assert os.path.dirname(item["A_paths"][0]) == str(root / "single" / "train")

    # the device pipeline (upload uint8 -> sgan_image_resize -> sgan_image_prep) against Pillow's on the same draws, bit for bit
    import random
    import image_prep as IP
    from PIL import Image
    opt.serial_batches, opt.nThreads = True, 0
    ds = SingleFolderDataset(opt)
    random.seed(21)
    got = ds[2]["A"][0].cpu().numpy()
    random.seed(21)
    x0, y0 = random.randint(0, 143 - 128), random.randint(0, 143 - 128)
    flip, rot = random.random() < 0.5, random.randint(0, 3)
    ref_img = np.asarray(Image.open(ds.paths[2]).convert("RGB").resize((143, 143), Image.BILINEAR), dtype=np.uint8)
    assert np.array_equal(got, IP.prep_pil(ref_img, x0, y0, 128, flip, rot))

    def epoch(threads):            # host decodes on worker threads must not change what an epoch yields
        import random
        opt.nThreads = threads
        random.seed(11)
        return [(d["A_paths"][0], d["A"].clone()) for d in SingleFolderDataset(opt)]
    serial, threaded = epoch(0), epoch(3)
    assert len(serial) == 5 and [p for p, _ in serial] == [p for p, _ in threaded]
    assert all(torch.equal(x, y) for (_, x), (_, y) in zip(serial, threaded))

    _write_images(str(root / "pairs" / "train"), 4, 300, 150, 2)
    _write_images(str(root / "pairs" / "test"), 2, 300, 150, 3)
    cnet = ["--name", "fold_cgan", "--model", "cgan", "--which_direction", "AtoB", "--dataset_mode", "aligned", "--fineSize", "128",
            "--loadSize", "140", "--which_model_netG", "unet_128", "--ngf", "8", "--which_channel", "rg_b", "--dataroot", str(root / "pairs")]
    m = train_driver.main(cnet + common + ["--which_model_netD", "n_layers", "--n_layers_D", "3", "--ndf", "8", "--scale_factor", "1",
                                          "--lambda_D", "1.0", "--no_lsgan", "--max_steps", "3"])
    assert all(np.isfinite(v) for v in m.get_current_errors().values())
    from supervised_gan_amd.data import AlignedFolderDataset
    copt = TrainOptions().parse(cnet + common + ["--which_model_netD", "n_layers", "--n_layers_D", "3", "--ndf", "8", "--scale_factor", "1",
                                                 "--lambda_D", "1.0", "--no_lsgan"], save=False)
    ads = AlignedFolderDataset(copt)
    random.seed(5)
    pair = ads[1]
    random.seed(5)
    wo, ho = random.randint(0, 140 - 128 - 1), random.randint(0, 140 - 128 - 1)
    aflip = random.random() < 0.5
    AB = np.asarray(Image.open(ads.paths[1]).convert("RGB").resize((280, 140), Image.BICUBIC), dtype=np.uint8)    # aligned_dataset.py:25
    assert np.array_equal(pair["A"][0].cpu().numpy(), IP.prep_pil(AB, wo, ho, 128, aflip, 0))
    assert np.array_equal(pair["B"][0].cpu().numpy(), IP.prep_pil(AB, 140 + wo, ho, 128, aflip, 0))
    out = test_driver.main(cnet + ["--gpu_ids", "0", "--checkpoints_dir", str(tmp_path / "ckpt"), "--norm", "instance",
                                   "--results_dir", str(tmp_path / "res"), "--how_many", "2"])
    assert len(out) == 4 and all(os.path.exists(p) and os.path.dirname(p).startswith(str(tmp_path / "res")) for p in out)

    _write_images(str(root / "unpaired" / "trainA"), 3, 150, 150, 4)
    _write_images(str(root / "unpaired" / "trainB"), 2, 160, 150, 5)
    unet = ["--name", "fold_cgan2", "--model", "cgan2", "--dataset_mode", "unaligned", "--fineSize", "128", "--loadSize", "140",
            "--which_model_netG", "unet_128", "--ngf", "8", "--which_channel", "rg_b", "--dataroot", str(root / "unpaired"),
            "--which_model_netD", "n_layers", "--n_layers_D", "3", "--ndf", "8", "--scale_factor", "1", "--lambda_D", "1.0", "--no_lsgan",
            "--max_steps", "3"]
    m = train_driver.main(unet + common)
    assert all(np.isfinite(v) for v in m.get_current_errors().values())
