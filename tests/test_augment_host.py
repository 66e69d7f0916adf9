"""CPU: host logic of the augmentation path -- the ctypes mirror of UnetAugmentRecipe has the C layout, the recipe makes
the reference's draws in the reference's order, the C ABI rejects malformed recipes without touching a GPU, and the numpy
restatement (oracle/augment_ref.py) gives the known answers its definitions imply."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import unet_studio_amd as U
from unet_studio_amd import augment as G
from oracle import augment_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ctypes_recipe_has_the_c_layout(tmp_path):
    fields = [f[0] for f in G.Recipe._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "unet_augment.h"\nint main(void){\n'
                   'printf("%zu\\n", sizeof(UnetAugmentRecipe));\n' +
                   "".join('printf("%%zu\\n", offsetof(UnetAugmentRecipe, %s));\n' % f for f in fields) + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert out[0] == C.sizeof(G.Recipe)
    assert out[1:] == [getattr(G.Recipe, f).offset for f in fields]


class Scripted:
    def __init__(self, values):
        self.values, self.draws = list(values), 0

    def __call__(self):
        v = self.values[self.draws % len(self.values)]
        self.draws += 1
        return np.float32(v)


def _all(level, **over):
    o = dict(G.DEFAULT_OPTIONS)
    for k in ("cropping", "truncation_z", "downsample_x", "downsample_y", "downsample_z", "noise", "ambient", "diffuse", "specular",
              "distortion", "zero_background", "rubber_stamping", "perlin_texture"):
        o[k] = level
    o.update(over)
    return o


def test_draw_counts_follow_the_reference_sequence():
    # everything off: only the view draws, .cu:385-415: resolution 1 + translation 3 + rotation 3 + aspect 3 + perspective 3 + lens 1
    one = Scripted([0.25])
    r = G.make_recipe(_all(0), (32, 32, 32), 1, True, 0, one=one)
    assert one.draws == 14 and not (r["crop"] or r["noise"] or r["rubber"] or r["perlin"] or r["zero_background"])
    # lens_distortion == 0 skips its draw (.cu:414)
    one = Scripted([0.25])
    G.make_recipe(_all(0, lens_distortion=0.0), (32, 32, 32), 1, True, 0, one=one)
    assert one.draws == 13
    # a 25 %..75 % switch costs one draw each (.cu:299-307); "On" (4) and "Off" (0) cost none
    one = Scripted([0.9])   # |0.9| is above every threshold: every probabilistic stage stays off
    r = G.make_recipe(_all(2), (32, 32, 32), 1, True, 0, one=one)
    #   3 downsample + crop + trunc + noise + ambient + diffuse + specular = 9, view 14, distortion 1, zero_bg 1, rubber 1, perlin 1
    assert one.draws == 9 + 14 + 4 and r["n_foci"] == 0
    # everything on: crop 1+1+3, trunc 2, ambient 1, diffuse 3, specular 3, view 14, foci 1 + n*(3+1+1),
    # stamps 5*9 + channels*5, perlin 2
    one = Scripted([0.5])
    r = G.make_recipe(_all(4, zero_background=0), (32, 32, 32), 2, True, 0, one=one)
    n = r["n_foci"]
    assert n == int(np.float32(0.5) * 3 * 0.5 + 5 * 0.5)
    assert one.draws == 5 + 2 + 1 + 3 + 3 + 14 + 1 + 5 * n + 45 + 10 + 2
    assert r["downsample"] == 1 and r["low_dims"] == [16, 16, 16]
    # not a label volume: the background stage draws nothing (.cu:449)
    one = Scripted([0.5])
    G.make_recipe(_all(4, zero_background=0), (32, 32, 32), 2, False, 0, one=one)
    assert one.draws == 5 + 2 + 1 + 3 + 3 + 14 + 1 + 5 * n


def test_recipe_is_a_function_of_the_seed():
    a = G.make_recipe(None, (32, 24, 16), 2, True, 7)
    b = G.make_recipe(None, (32, 24, 16), 2, True, 7)
    c = G.make_recipe(None, (32, 24, 16), 2, True, 8)
    assert bytes(G.to_struct(a)) == bytes(G.to_struct(b)) != bytes(G.to_struct(c))
    assert sorted(a["perm"].tolist()) == sorted((np.arange(512) & 255).tolist()) or not a["perlin"]


def test_c_abi_rejects_malformed_recipes_without_a_gpu():
    r = G.to_struct(G.make_recipe(None, (8, 8, 8), 1, True, 0))
    n = C.c_size_t()
    assert U.engine.lib.unet_augment_scratch_bytes(C.byref(r), C.byref(n)) == 0 and n.value > 4 * 8 * 8 * 8 * 2
    for field, bad in (("channels", 0), ("channels", 9), ("n_foci", 17), ("trunc_top", -1)):
        q = G.to_struct(G.make_recipe(None, (8, 8, 8), 1, True, 0))
        setattr(q, field, bad)
        assert U.engine.lib.unet_augment_scratch_bytes(C.byref(q), C.byref(n)) != 0
        assert b"unet_augment" in U.engine.lib.unet_last_error()
    q = G.to_struct(G.make_recipe(None, (8, 8, 8), 1, True, 0))
    q.dims[1] = 0
    assert U.engine.lib.unet_augment_scratch_bytes(C.byref(q), C.byref(n)) != 0
    with pytest.raises(ValueError):
        G.make_recipe(None, (8, 8, 8), 9, True, 0)


def _identity(shape, channels, is_label=True):
    r = G.make_recipe(_all(0, lens_distortion=0.0, perspective=0.0), shape, channels, is_label, 0)
    r["view"] = (np.eye(3, dtype=np.float32).reshape(9), np.zeros(3, np.float32))
    return r


def test_oracle_known_answers():
    W, H, D = 10, 8, 6
    rs = np.random.RandomState(0)
    img = rs.rand(1, D, H, W).astype(np.float32) - 0.2
    lab = (rs.rand(D, H, W) * 3).astype(np.int64).astype(np.float32)
    r = _identity((W, H, D), 1)
    o, l = R.augment(r, img, lab)
    assert np.array_equal(l, lab)
    assert np.array_equal(o[0], np.maximum(img[0], 0) / img[0].max())     # lower_threshold + normalize, .cu:449-452
    # half-voxel shift along x: linear interpolation of neighbours, majority keeps the first corner on a tie
    r["view"] = (np.eye(3, dtype=np.float32).reshape(9), np.array([0.5, 0, 0], np.float32))
    o, l = R.augment(r, np.abs(img), lab)
    a = np.abs(img[0])
    mid = (a[:, :, :-1] + np.float32(0.5) * (a[:, :, 1:] - a[:, :, :-1]))
    assert np.allclose(o[0][:, :, :-1] * o[0].max() * mid.max() / o[0].max(), mid, atol=1e-6)
    assert np.array_equal(o[0][:, :, -1], np.zeros((D, H), np.float32))     # x = 9.5 is outside
    assert np.array_equal(l[:, :, :-1], lab[:, :, :-1])
    # zero_background keeps labelled voxels only (.cu:452-457)
    r = _identity((W, H, D), 1)
    r["zero_background"] = 1
    o, l = R.augment(r, np.abs(img), lab)
    assert np.all(o[0][lab == 0] == 0) and np.all(o[0][lab != 0] > 0)
    # truncation: slices cleared in image and label (.cu:31-59)
    r = _identity((W, H, D), 1)
    r["trunc_top"], r["trunc_bottom"] = 1, 2
    o, l = R.augment(r, np.abs(img), lab)
    assert not l[:2].any() and not l[-1:].any() and not o[0][:2].any() and not o[0][-1:].any() and l[2:-1].any()
    # cropping only where a label is, first channel only, label cleared (.cu:6-24, 339-340)
    r = _identity((W, H, D), 2)
    r["crop"], r["crop_pos"], r["crop_radius"], r["crop_value"] = 1, [5, 4, 3], np.float32(2.0), np.float32(7.0)
    img2 = np.abs(rs.rand(2, D, H, W).astype(np.float32))
    full = np.ones((D, H, W), np.float32)
    o, l = R.augment(r, img2, full)
    assert l[3, 4, 5] == 0 and l[3, 4, 7] == 0 and l[3, 4, 8] == 1 and l[0, 0, 0] == 1
    assert o[0][3, 4, 5] == 1.0 and np.array_equal(o[1], img2[1] / img2[1].max())
    # the noise stream is U(0,1] and depends on the seed
    u = R._hash_u01(np.arange(100000, dtype=np.uint64), 3)
    assert 0 < u.min() and u.max() <= 1 and abs(u.mean() - 0.5) < 0.01 and not np.array_equal(u, R._hash_u01(np.arange(100000, dtype=np.uint64), 4))


def test_oracle_perlin_is_a_lattice_noise():
    p = (np.arange(512) & 255).astype(np.int64)
    np.random.RandomState(1).shuffle(p)
    # zero at lattice points (gradient noise), bounded, continuous
    g = np.arange(5, dtype=np.float32)
    assert np.all(R._perlin_at(p, g, g[::-1].copy(), g) == 0)
    x = np.linspace(0, 4, 401, dtype=np.float32)
    v = R._perlin_at(p, x, x * np.float32(0.7), x * np.float32(1.3))
    assert np.abs(v).max() <= 1.5 and np.abs(np.diff(v)).max() < 0.05


# ---- simulate_modality (train.cpp:43-178) ----
def test_simulate_recipe_layout_and_draws(tmp_path):
    fields = [f[0] for f in G.SimRecipe._fields_]
    src = tmp_path / "layout_sim.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "unet_augment.h"\nint main(void){\n'
                   'printf("%zu\\n", sizeof(UnetSimulateRecipe));\n' +
                   "".join('printf("%%zu\\n", offsetof(UnetSimulateRecipe, %s));\n' % f for f in fields) + "return 0;}\n")
    exe = tmp_path / "layout_sim"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert out[0] == C.sizeof(G.SimRecipe) and out[1:] == [getattr(G.SimRecipe, f).offset for f in fields]
    # draw order (train.cpp:52-80): lut from the float generator first, then per term (a,b) until a+b != 0, c, d from the integer
    # generator and w from the float one, then gamma
    ints = iter([0, 0, 0, 0, 3, 1, 2, 0] + [1, 2, 3, 0] * 19)
    floats = iter([0.0, 0.5, 1.0] + [0.25] * 20 + [0.5])
    r = G.make_simulate_recipe((8, 8, 8), 2, 0, rand_int=lambda n: next(ints), rand_float=lambda: np.float32(next(floats)))
    assert [float(v) for v in r["lut"]] == [np.float32(0.4), np.float32(0.5), np.float32(0.4) + np.float32(0.2)]
    assert r["terms"][0][:4] == (3, 1, 2, 0) and r["terms"][1][:4] == (1, 2, 3, 0) and len(r["terms"]) == 20
    assert r["gamma"] == np.float32(0.6) + np.float32(1.2) * np.float32(0.5)
    assert G.make_simulate_recipe((8, 8, 8), None, 0)["with_label"] == 0
    n = C.c_size_t()
    s = G.sim_to_struct(r)
    assert U.engine.lib.unet_simulate_modality_scratch_bytes(C.byref(s), C.byref(n)) == 0 and n.value > 2 * 4 * 512
    s.term_a[0] = 4
    assert U.engine.lib.unet_simulate_modality_scratch_bytes(C.byref(s), C.byref(n)) != 0
    with pytest.raises(ValueError):
        G.make_simulate_recipe((8, 8, 8), 256, 0)


def test_oracle_simulate_modality_known_answers():
    # one monomial x^1 with weight 1, gamma 1, no labels: out = stretch(x) over the voxels above the 0.02 cut
    r = {"dims": [4, 4, 4], "with_label": 0, "max_label": 0, "lut": [], "gamma": np.float32(1.0),
         "terms": [(1, 0, 0, 0, np.float32(1.0))] + [(1, 0, 0, 0, np.float32(0.0))] * 19}
    x = np.linspace(0.0, 1.0, 64, dtype=np.float32).reshape(4, 4, 4)
    o = R.simulate_modality(r, x)
    keep = x > 0.02
    mn, mx = x[keep].min(), x[keep].max()
    assert np.allclose(o[keep], (x[keep] - mn) / (mx - mn), atol=1e-6) and np.all(o[~keep] == 0)
    # the smoothing: a constant stays constant, an impulse spreads as (1,2,1)^3/64 with replicated borders
    assert np.array_equal(R._smooth(np.full((3, 4, 5), 2.0, np.float32)), np.full((3, 4, 5), 2.0, np.float32))
    imp = np.zeros((5, 5, 5), np.float32); imp[2, 2, 2] = 64
    sm = R._smooth(imp)
    assert sm[2, 2, 2] == 8 and sm[2, 2, 1] == 4 and sm[1, 1, 2] == 2 and sm[1, 1, 1] == 1 and sm.sum() == 64
    # with labels: the stretch uses labelled voxels only, the look-up table sets the tissue level
    r2 = dict(r, with_label=1, max_label=1, lut=[np.float32(0.4), np.float32(0.6)])
    lab = np.zeros((4, 4, 4), np.float32); lab[2:] = 1
    o2 = R.simulate_modality(r2, x, lab)
    sel = keep & (lab != 0)
    assert o2[sel].min() == 0.0 and o2[sel].max() == 1.0 and np.all(o2[keep & (lab == 0)] == 0.0)   # below the labelled minimum: clamped
