"""GPU: the on-GPU augmentation (include/unet_augment.h, through the C ABI) against the numpy restatement of the
reference's kernels (oracle/augment_ref.py) on the same recipes, then size-independent properties at config 5's size
(2 channels of 256^3)."""
import numpy as np
import pytest
import torch

import unet_studio_amd as U
from unet_studio_amd import augment as G
from oracle import augment_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sample(shape, channels, classes, seed):
    W, H, D = shape
    rs = np.random.RandomState(seed)
    z, y, x = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    # an "organ": nested ellipsoids of classes, so that labels have interiors and a background
    rr = ((x - W / 2) / (W * 0.35)) ** 2 + ((y - H / 2) / (H * 0.3)) ** 2 + ((z - D / 2) / (D * 0.4)) ** 2
    lab = np.zeros((D, H, W), np.float32)
    for k in range(1, classes):
        lab[rr < (1.0 - (k - 1) / classes) ** 2] = k
    img = (rs.rand(channels, D, H, W) * 0.3 + (lab[None] > 0) * 0.5 + lab[None] * 0.05).astype(np.float32)
    return img, lab


def _options(level, **over):
    """level 4 = every stage always on (no apply() draws), 0 = off."""
    o = dict(G.DEFAULT_OPTIONS)
    for k in ("cropping", "truncation_z", "downsample_x", "downsample_y", "downsample_z", "noise", "ambient", "diffuse", "specular",
              "distortion", "zero_background", "rubber_stamping", "perlin_texture"):
        o[k] = level
    o.update(over)
    return o


def _run(recipe, img, lab):
    x = torch.from_numpy(img.copy()).to(DEV)
    t = torch.from_numpy(lab.copy()).to(DEV)
    G.augment(recipe, x, t)
    torch.cuda.synchronize()
    return x.cpu().numpy(), t.cpu().numpy()


def _compare(recipe, img, lab, is_label=True):
    got_i, got_l = _run(recipe, img, lab)
    ref_i, ref_l = R.augment(recipe, img, lab)
    assert np.isfinite(got_i).all() and np.isfinite(got_l).all()
    assert (ref_i > 0).mean() > 0.05 and ref_i.max() == 1.0, "degenerate case: nothing left in view"
    if is_label:
        # positions pass through sinf/cosf, whose last bit differs between the GPU's and numpy's libm: a vote that is an exact
        # tie flips.  Those voxels are excluded from the image comparison (the blend is gated on the label) and must be rare.
        same = got_l == ref_l
        assert same.mean() > 0.999, "label mismatch fraction %.5f" % (1 - same.mean())
    else:
        same = np.ones(ref_l.shape, bool)
        assert np.abs(got_l - ref_l).max() < 5e-6
    err = np.abs(got_i - ref_i)[:, same].max()
    assert err < 5e-6, err      # images are in [0,1] after the normalisation; observed worst 4.2e-7 (profiles/augment_label_agreement.py)
    return got_i, got_l


CASES = [
    # (shape (W,H,D), channels, options, is_label)
    ((24, 20, 16), 1, _options(0, lens_distortion=0.0, perspective=0.0), True),                 # view only
    ((24, 20, 16), 2, _options(4, zero_background=0), True),                                     # everything, with blending
    ((24, 20, 16), 2, _options(4), True),                                                        # zero_background stops early
    ((33, 17, 21), 1, _options(4, zero_background=0, rubber_stamping=0), True),                  # Perlin only, odd sizes
    ((33, 17, 21), 2, _options(4, zero_background=0, perlin_texture=0), True),                   # stamps only
    ((20, 28, 12), 2, _options(4), False),                                                       # label is an image: linear, no background stage
    ((32, 32, 32), 2, dict(G.DEFAULT_OPTIONS), True),                                            # the shipped option set (probabilistic stages)
    ((16, 16, 40), 3, _options(4, zero_background=0, downsample_x=0, downsample_y=0), True),     # z-only resolution loss
]


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_augmentation_matches_the_restated_kernels(case, seed):
    shape, channels, opt, is_label = CASES[case]
    img, lab = _sample(shape, channels, 4, seed)
    r = G.make_recipe(opt, shape, channels, is_label, seed * 7919 + case)
    _compare(r, img, lab, is_label)


def test_each_stage_alone():
    shape, img_lab = (28, 24, 20), None
    img, lab = _sample(shape, 2, 3, 0)
    base = _options(0, lens_distortion=0.0, perspective=0.0)
    for k in ("cropping", "truncation_z", "downsample_x", "noise", "ambient", "diffuse", "specular", "distortion", "rubber_stamping",
              "perlin_texture", "zero_background"):
        o = dict(base)
        o[k] = 4
        if k == "distortion":
            o["lens_distortion"] = 0.1
        r = G.make_recipe(o, shape, 2, True, 11)
        if k == "cropping":   # the reference's own location draw lands outside the volume (.cu:338): put one inside as well
            r["crop_pos"], r["crop_radius"] = [14, 12, 10], np.float32(6.5)
        _compare(r, img, lab)


def test_identity_recipe_is_a_normalisation():
    shape = (20, 16, 12)
    img, lab = _sample(shape, 2, 3, 4)
    r = G.make_recipe(_options(0, lens_distortion=0.0, perspective=0.0), shape, 2, True, 0)
    r["view"] = (np.eye(3, dtype=np.float32).reshape(9), np.zeros(3, np.float32))
    got_i, got_l = _run(r, img, lab)
    assert np.array_equal(got_l, lab)
    for c in range(2):
        assert np.array_equal(got_i[c], img[c] / img[c].max())
    # an integer shift moves voxels exactly
    r["view"] = (np.eye(3, dtype=np.float32).reshape(9), np.array([2, -1, 3], np.float32))
    got_i, got_l = _run(r, img, lab)
    exp = np.zeros_like(lab)
    exp[0:9, 1:16, 0:18] = lab[3:12, 0:15, 2:20]     # out[z,y,x] = in[z+3, y-1, x+2]
    assert np.array_equal(got_l, exp)


def test_rejects_bad_input():
    shape = (8, 8, 8)
    r = G.make_recipe(None, shape, 1, True, 0)
    x = torch.zeros(8 * 8 * 8, device=DEV)
    with pytest.raises(U.UNetError):
        G.augment(r, x, torch.zeros(7, device=DEV))
    with pytest.raises(U.UNetError):
        G.augment(r, x.cpu(), x.cpu())
    with pytest.raises(U.UNetError):
        G.augment(r, x.double(), x.double())


def test_config5_size_properties():
    """2 channels of 256^3 (BASELINE.json configs[4]): too large for the numpy restatement, so size-independent properties."""
    n, ch = 256, 2
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    x = torch.rand((ch, n, n, n), device=DEV, generator=g)
    zz = torch.arange(n, device=DEV, dtype=torch.float32)
    rr = ((zz[:, None, None] - 128) / 90) ** 2 + ((zz[None, :, None] - 128) / 80) ** 2 + ((zz[None, None, :] - 128) / 100) ** 2
    lab = (rr < 1).float() + (rr < 0.5).float() + (rr < 0.2).float()
    x += (lab > 0)[None] * 0.5
    r = G.make_recipe(_options(4, zero_background=0), (n, n, n), ch, True, 123)
    x1, l1 = x.clone(), lab.clone()
    G.augment(r, x1.view(-1), l1.view(-1))
    x2, l2 = x.clone(), lab.clone()
    G.augment(r, x2.view(-1), l2.view(-1))
    torch.cuda.synchronize()
    assert torch.equal(x1, x2) and torch.equal(l1, l2)          # deterministic (the maxima are order-independent)
    assert torch.isfinite(x1).all()
    for c in range(ch):
        assert float(x1[c].max()) == 1.0 and float(x1[c].min()) >= 0.0
    assert set(torch.unique(l1).tolist()) <= {0.0, 1.0, 2.0, 3.0}  # majority resampling invents no class
    assert 0.02 < float((l1 > 0).float().mean()) < 0.6
    # zero_background: nothing but the labelled voxels survives
    r0 = G.make_recipe(_options(4), (n, n, n), ch, True, 123)
    x3, l3 = x.clone(), lab.clone()
    G.augment(r0, x3.view(-1), l3.view(-1))
    assert torch.equal(l3, l1)
    assert float(x3[:, l3 == 0].abs().max()) == 0.0


def test_augmented_source_feeds_a_train_step():
    arch = ("conv8,ks3,stride1+norm,leaky_relu+conv8,ks3,stride1+norm,leaky_relu\n"
            "conv16,ks3,stride2+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu+conv_trans8,ks2,stride2\n"
            "conv8,ks3,stride1+norm,leaky_relu+conv8,ks3,stride1+norm,leaky_relu+conv4,ks1,stride1")
    m = U.UNet3d(2, 4, arch, device=DEV, dtype="bf16", seed=0)
    src = U.AugmentedVolumes(U.SyntheticVolumes(2, 4, (16, 24, 32), DEV, cache=2))
    x, t = src(0)
    assert x.shape == (1, 2, 16, 24, 32) and t.shape == (1, 16, 24, 32) and t.dtype == torch.int64
    assert float(x.max()) <= 1.0 and int(t.max()) <= 3
    xa, ta = src(0)
    assert torch.equal(x, xa) and torch.equal(t, ta)            # seed = sample index
    tr = U.Trainer(m, U.TrainingParam(batch_size=2, epoch=10, learning_rate=0.01), lambda i: src(i % 2))
    s = tr.step()
    torch.cuda.synchronize()
    assert torch.isfinite(s).all()


def test_sample_ring_hands_over_the_synchronous_samples():
    """PrefetchedVolumes (the device-resident form of the reader ring, train.cpp:259-486): sample i+1 is produced on a side stream
    while step i trains.  The samples, and therefore the trained parameters, are the synchronous source's bit for bit."""
    arch = ("conv8,ks3,stride1+norm,leaky_relu\nconv16,ks3,stride2+norm,leaky_relu+conv_trans8,ks2,stride2\n"
            "conv8,ks3,stride1+norm,leaky_relu+conv4,ks1,stride1")
    base = U.SyntheticVolumes(2, 4, (16, 24, 32), DEV, cache=4)
    direct = U.AugmentedVolumes(lambda i: base(i % 4))
    ring = U.PrefetchedVolumes(U.AugmentedVolumes(lambda i: base(i % 4)), stride=1)
    for i in (0, 1, 2, 5, 6):                        # consecutive requests are prefetched, a jump is produced on demand
        xa, ta = direct(i)
        xb, tb = ring(i)
        torch.cuda.synchronize()
        assert torch.equal(xa, xb) and torch.equal(ta, tb), i

    def run(feed):
        m = U.UNet3d(2, 4, arch, device=DEV, dtype="bf16", seed=0)
        tr = U.Trainer(m, U.TrainingParam(batch_size=2, epoch=10, learning_rate=0.01), feed)
        for _ in range(4):
            tr.step()
        torch.cuda.synchronize()
        return m.flat_params.clone()

    pa = run(U.AugmentedVolumes(lambda i: base(i % 4)))
    pb = run(U.PrefetchedVolumes(U.AugmentedVolumes(lambda i: base(i % 4)), stride=1))
    assert torch.equal(pa, pb)


# ---- simulate_modality (train.cpp:43-178) ----
@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("with_label", [True, False])
@pytest.mark.parametrize("shape", [(24, 20, 16), (33, 17, 21), (8, 8, 8)])
def test_simulate_modality_matches_the_restatement(shape, with_label, seed):
    img, lab = _sample(shape, 1, 5, seed)
    t1w = np.clip(img[0], 0, 1).astype(np.float32)
    t1w[lab == 0] *= 0.02          # a background at / below the 0.02 cut (train.cpp:86-90)
    r = G.make_simulate_recipe(shape, 4 if with_label else None, seed)
    x = torch.from_numpy(t1w.copy()).to(DEV)
    l = torch.from_numpy(lab.copy()).to(DEV)
    G.simulate(r, x, l if with_label else None)
    torch.cuda.synchronize()
    got, ref = x.cpu().numpy(), R.simulate_modality(r, t1w, lab)
    assert np.isfinite(got).all() and got.min() >= 0.0 and got.max() <= 1.0
    assert (ref > 0).mean() > 0.1, "degenerate case"
    assert np.abs(got - ref).max() < 2e-5     # powf / the min-max stretch amplify last-bit differences of the 20-term sum
    assert np.all(got[t1w <= 0.02] == 0.0)    # the cut voxels stay 0 through the stretch (clamped)


def test_simulate_modality_at_full_size_and_errors():
    n = 256
    g = torch.Generator(device=DEV)
    g.manual_seed(1)
    x = torch.rand((n, n, n), device=DEV, generator=g)
    lab = (torch.rand((n, n, n), device=DEV, generator=g) * 6).floor()
    r = G.make_simulate_recipe((n, n, n), 5, 7)
    a, b = x.clone(), x.clone()
    G.simulate(r, a, lab)
    G.simulate(r, b, lab)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and float(a.min()) == 0.0 and float(a.max()) == 1.0     # deterministic, stretched over labelled voxels
    with pytest.raises(U.UNetError):
        G.simulate(r, x.clone(), None)
    with pytest.raises(U.UNetError):
        G.simulate(r, x.cpu(), lab.cpu())
