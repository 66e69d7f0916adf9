"""Host side of the on-GPU sample augmentation (include/unet_augment.h): mirrors
`visual_perception_augmentation(options, input, label, is_label, image_shape, random_seed)`
(visual_perception_augmentation.cpp:163-170, GPU path visual_perception_augmentation.cu:282-544).

The reference draws its random numbers between kernel launches; here `make_recipe` makes every draw first, in the
reference's order (each `apply()` / `range()` / `one()` call of .cu:297-513 has one line below), and the engine then runs
a deterministic function of the recipe.  What TIPL defines and the reference tree does not contain is OUR choice, stated
where it is made (parity unpinned for exactly these): the uniform generator (`tipl::uniform_dist`), how an
`affine_param` becomes a matrix (`tipl::transformation_matrix`), `std::shuffle`'s permutation.  A reference-side caller
fills the recipe from its own TIPL objects instead.

No compute happens in this file and nothing falls back to the CPU: `augment()` needs libunet_hip.so and device tensors.
"""
import ctypes as C
import math

import numpy as np

from . import engine

MAX_CHANNELS, MAX_FOCI, STAMPS = 8, 16, 5

# name -> default: the fifth column of options.txt:1-39 (train.cpp:1167 loads them into param.options)
DEFAULT_OPTIONS = {
    "cropping": 0, "cropping_size_min": 0.1, "cropping_size_max": 0.2, "truncation_z": 1,
    "downsample_x": 2, "downsample_x_ratio": 0.5, "downsample_y": 2, "downsample_y_ratio": 0.5,
    "downsample_z": 2, "downsample_z_ratio": 0.5, "noise": 2, "noise_mag": 0.2,
    "ambient": 2, "ambient_mag": 2.0, "diffuse": 2, "diffuse_mag": 2.0,
    "specular": 2, "specular_freq": 2.0, "specular_mag": 0.5,
    "translocation_ratio": 0.2, "rotation_x": 0.2, "rotation_y": 0.2, "rotation_z": 0.2,
    "scaling_up": 1.25, "scaling_down": 0.8, "aspect_ratio": 1.25, "perspective": 0.1, "lens_distortion": 0.1,
    "distortion": 1, "distortion_count": 3, "distortion_radius_min": 0.1, "distortion_radius_max": 0.5,
    "distortion_mag_min": 0.05, "distortion_mag_max": 0.1,
    "zero_background": 1, "rubber_stamping": 2, "rubber_stamping_mag": 0.5, "perlin_texture": 2, "perlin_texture_mag": 0.5,
}

F = np.float32


class Affine(C.Structure):
    _fields_ = [("sr", C.c_float * 9), ("shift", C.c_float * 3)]


class Recipe(C.Structure):
    """UnetAugmentRecipe, field for field."""
    _fields_ = [
        ("dims", C.c_int * 3), ("channels", C.c_int), ("is_label", C.c_int),
        ("downsample", C.c_int), ("low_dims", C.c_int * 3),
        ("crop", C.c_int), ("crop_pos", C.c_int * 3), ("crop_radius", C.c_float), ("crop_value", C.c_float),
        ("trunc_top", C.c_int), ("trunc_bottom", C.c_int),
        ("noise", C.c_int), ("noise_mag", C.c_float), ("noise_seed", C.c_uint),
        ("ambient", C.c_int), ("ambient_value", C.c_float),
        ("diffuse", C.c_int), ("diffuse_dir", C.c_float * 3), ("diffuse_mag", C.c_float),
        ("specular", C.c_int), ("specular_pos", C.c_int * 3), ("specular_freq", C.c_float), ("specular_mag", C.c_float),
        ("view", Affine), ("has_perspective", C.c_int), ("perspective", C.c_float * 3),
        ("has_lens", C.c_int), ("lens_magnitude", C.c_float),
        ("n_foci", C.c_int), ("foci_pos", (C.c_int * 3) * MAX_FOCI), ("foci_radius", C.c_float * MAX_FOCI),
        ("foci_magnitude", C.c_float * MAX_FOCI),
        ("zero_background", C.c_int), ("rubber", C.c_int), ("stamp", Affine * STAMPS),
        ("stamp_mag", (C.c_float * STAMPS) * MAX_CHANNELS),
        ("perlin", C.c_int), ("perm", C.c_ubyte * 512), ("perlin_zoom", C.c_float), ("perlin_mag", C.c_float),
    ]


engine._sig("unet_augment_scratch_bytes", C.c_int, C.POINTER(Recipe), C.POINTER(C.c_size_t))
engine._sig("unet_augment_run", C.c_int, C.POINTER(Recipe), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
EXPORTS = ["unet_augment_scratch_bytes", "unet_augment_run"]


class UniformDist:
    """Stand-in for tipl::uniform_dist<float>(-1, 1, seed) (TIPL; .cu:297): MT19937 seeded with `seed`, one fp32 per call."""

    def __init__(self, seed):
        self._rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
        self.draws = 0

    def __call__(self):
        self.draws += 1
        return F(self._rs.uniform(-1.0, 1.0))


def affine_matrix(translation, rotation, scaling, shape):
    """Stand-in for tipl::transformation_matrix<float>(affine_param, shape, (1,1,1), shape, (1,1,1)) (TIPL; .cu:402,470):
    pos' = R * diag(scaling) * (pos - c) + c + translation, c = shape / 2, R = Rx(rx) * Ry(ry) * Rz(rz) (radians).
    Returns (sr[9] row-major, shift[3]) as float32."""
    rx, ry, rz = (float(v) for v in rotation)
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    M = Rx @ Ry @ Rz @ np.diag([float(v) for v in scaling])
    c = np.array([float(s) for s in shape]) / 2.0
    shift = c - M @ c + np.array([float(v) for v in translation])
    return M.astype(F).reshape(9), shift.astype(F)


def make_recipe(options, image_shape, channels, is_label, random_seed, label_depth=None, one=None):
    """Every draw of visual_perception_augmentation_cuda, in its order -> dict with the fields of UnetAugmentRecipe.
    image_shape = (width, height, depth).  `one` may be any callable returning U(-1,1) floats (default UniformDist)."""
    opt = dict(DEFAULT_OPTIONS)
    opt.update(options or {})
    W, H, D = (int(v) for v in image_shape)
    if not 1 <= channels <= MAX_CHANNELS:
        raise ValueError("channels must be in 1..%d" % MAX_CHANNELS)
    one = one or UniformDist(random_seed)

    def rng(lo, hi):   # `range`, .cu:298
        return F(F(one() * F(F(hi) - F(lo))) * F(0.5) + F(F(hi) + F(lo)) * F(0.5))

    def apply(name):   # .cu:299-307
        index = int(opt[name])
        if index == 0:
            return False
        if index >= 4:
            return True
        return abs(one()) < F(index) * F(0.25)

    def location(lo, hi):   # random_location, .cu:309-310: truncation toward zero as the int vector constructor does
        return [int(F(W - 1) * rng(lo, hi)), int(F(H - 1) * rng(lo, hi)), int(F(D - 1) * rng(lo, hi))]

    r = {"dims": [W, H, D], "channels": int(channels), "is_label": int(bool(is_label))}

    dx, dy, dz = apply("downsample_x"), apply("downsample_y"), apply("downsample_z")   # .cu:322-324
    r["downsample"] = int(dx or dy or dz)
    r["low_dims"] = [max(1, int(F(W) * F(opt["downsample_x_ratio"] if dx else 1.0))),
                     max(1, int(F(H) * F(opt["downsample_y_ratio"] if dy else 1.0))),
                     max(1, int(F(D) * F(opt["downsample_z_ratio"] if dz else 1.0)))]

    r["crop"] = int(apply("cropping"))   # .cu:333-341
    r["crop_pos"], r["crop_radius"], r["crop_value"] = [0, 0, 0], F(0), F(0)
    if r["crop"]:
        size = F(rng(opt["cropping_size_min"], opt["cropping_size_max"]) * F(W))
        r["crop_value"] = rng(0.0, 2.0)
        r["crop_pos"] = location(F(size), F(F(1.0) - size))   # as written at .cu:338 (a voxel count used as a fraction)
        r["crop_radius"] = size

    r["trunc_top"] = r["trunc_bottom"] = 0   # .cu:343-354; label.depth() is the label volume's own depth
    if apply("truncation_z"):
        depth = F(label_depth if label_depth is not None else D)
        r["trunc_top"] = min(D, int(abs(F(F(one() * F(0.5)) * depth))))
        r["trunc_bottom"] = min(D, int(abs(F(F(one() * F(0.5)) * depth))))

    r["noise"] = int(apply("noise"))   # .cu:356-361
    r["noise_mag"], r["noise_seed"] = F(opt["noise_mag"]), int(random_seed) & 0xFFFFFFFF
    r["ambient"] = int(apply("ambient"))   # .cu:363-368
    r["ambient_value"] = F(rng(0.0, 1.0) * F(opt["ambient_mag"])) if r["ambient"] else F(0)
    r["diffuse"] = int(apply("diffuse"))   # .cu:369-374
    r["diffuse_dir"] = [rng(-0.5, 0.5), rng(-0.5, 0.5), rng(-0.5, 0.5)] if r["diffuse"] else [F(0)] * 3
    r["diffuse_mag"] = F(opt["diffuse_mag"])
    r["specular"] = int(apply("specular"))   # .cu:375-380
    r["specular_pos"] = location(0.4, 0.6) if r["specular"] else [0, 0, 0]
    r["specular_freq"], r["specular_mag"] = F(opt["specular_freq"]), F(opt["specular_mag"])

    # .cu:385-401: the order of the initialiser list
    resolution = rng(F(1.0) / F(opt["scaling_up"]), F(1.0) / F(opt["scaling_down"]))
    tr = [F(one() * F(opt["translocation_ratio"])) * F(n) for n in (W, H, D)]
    rot = [F(one() * F(opt[k])) for k in ("rotation_x", "rotation_y", "rotation_z")]
    sc = [F(resolution * rng(F(1.0) / F(opt["aspect_ratio"]), opt["aspect_ratio"])) for _ in range(3)]
    r["view"] = affine_matrix(tr, rot, sc, (W, H, D))
    r["perspective"] = [F(F(rng(-0.5, 0.5) * F(opt["perspective"])) / F(n)) for n in (W, H, D)]   # .cu:405-407
    r["has_perspective"] = int(opt["perspective"] > 0.0)
    r["has_lens"] = int(opt["lens_distortion"] > 0.0)
    r["lens_magnitude"] = F(rng(0.0, 1.0) * F(opt["lens_distortion"])) if opt["lens_distortion"] != 0.0 else F(0)   # .cu:414-415
    r["n_foci"], r["foci_pos"], r["foci_radius"], r["foci_magnitude"] = 0, [], [], []
    if apply("distortion"):   # .cu:417-429
        num = int(rng(1.0, F(opt["distortion_count"]) + F(1.0)))
        for _ in range(max(0, min(num, MAX_FOCI))):
            r["foci_pos"].append(location(0.3, 0.7))
            r["foci_radius"].append(F(F(W) * rng(opt["distortion_radius_min"], opt["distortion_radius_max"])))
            r["foci_magnitude"].append(rng(opt["distortion_mag_min"], opt["distortion_mag_max"]))
        r["n_foci"] = len(r["foci_pos"])

    r["zero_background"] = r["rubber"] = r["perlin"] = 0
    r["stamp"], r["stamp_mag"] = [], [[F(0)] * STAMPS for _ in range(channels)]
    r["perm"], r["perlin_zoom"], r["perlin_mag"] = np.zeros(512, np.uint8), F(0), F(0)
    if is_label:   # .cu:449
        r["zero_background"] = int(apply("zero_background"))
        if not r["zero_background"]:
            r["rubber"] = int(apply("rubber_stamping"))
            if r["rubber"]:   # .cu:460-488
                pi2 = F(math.pi * 2.0)
                for _ in range(STAMPS):
                    t = [F(F(one() * F(n)) * F(0.5)) for n in (W, H, D)]
                    a = [F(one() * pi2) for _ in range(3)]
                    s = [rng(0.8, 1.25) for _ in range(3)]
                    r["stamp"].append(affine_matrix(t, a, s, (W, H, D)))
                for c in range(channels):
                    for k in range(STAMPS):
                        r["stamp_mag"][c][k] = F(rng(0.0, 1.0) * F(opt["rubber_stamping_mag"]))
            r["perlin"] = int(apply("perlin_texture"))
            if r["perlin"]:   # .cu:490-513; std::shuffle(p, mt19937(seed)) is the C++ library's permutation: ours is numpy's
                p = (np.arange(512) & 255).astype(np.uint8)
                np.random.RandomState(int(random_seed) & 0xFFFFFFFF).shuffle(p)
                r["perm"] = p
                r["perlin_zoom"] = rng(0.005, 0.05)
                r["perlin_mag"] = F(rng(0.0, 1.0) * F(opt["perlin_texture_mag"]))
    return r


def to_struct(r):
    s = Recipe()
    s.dims[:] = r["dims"]
    s.channels, s.is_label = r["channels"], r["is_label"]
    s.downsample = r["downsample"]
    s.low_dims[:] = r["low_dims"]
    s.crop, s.crop_radius, s.crop_value = r["crop"], float(r["crop_radius"]), float(r["crop_value"])
    s.crop_pos[:] = r["crop_pos"]
    s.trunc_top, s.trunc_bottom = r["trunc_top"], r["trunc_bottom"]
    s.noise, s.noise_mag, s.noise_seed = r["noise"], float(r["noise_mag"]), r["noise_seed"]
    s.ambient, s.ambient_value = r["ambient"], float(r["ambient_value"])
    s.diffuse, s.diffuse_mag = r["diffuse"], float(r["diffuse_mag"])
    s.diffuse_dir[:] = [float(v) for v in r["diffuse_dir"]]
    s.specular, s.specular_freq, s.specular_mag = r["specular"], float(r["specular_freq"]), float(r["specular_mag"])
    s.specular_pos[:] = r["specular_pos"]

    def put(dst, m):
        dst.sr[:] = [float(v) for v in m[0]]
        dst.shift[:] = [float(v) for v in m[1]]
    put(s.view, r["view"])
    s.has_perspective, s.has_lens, s.lens_magnitude = r["has_perspective"], r["has_lens"], float(r["lens_magnitude"])
    s.perspective[:] = [float(v) for v in r["perspective"]]
    s.n_foci = r["n_foci"]
    for k in range(r["n_foci"]):
        s.foci_pos[k][:] = r["foci_pos"][k]
        s.foci_radius[k], s.foci_magnitude[k] = float(r["foci_radius"][k]), float(r["foci_magnitude"][k])
    s.zero_background, s.rubber, s.perlin = r["zero_background"], r["rubber"], r["perlin"]
    for k, m in enumerate(r["stamp"]):
        put(s.stamp[k], m)
    for c in range(r["channels"]):
        for k in range(STAMPS):
            s.stamp_mag[c][k] = float(r["stamp_mag"][c][k])
    s.perm[:] = [int(v) for v in r["perm"]]
    s.perlin_zoom, s.perlin_mag = float(r["perlin_zoom"]), float(r["perlin_mag"])
    return s


def scratch_bytes(recipe):
    s = recipe if isinstance(recipe, Recipe) else to_struct(recipe)
    n = C.c_size_t()
    engine.check(engine.lib.unet_augment_scratch_bytes(C.byref(s), C.byref(n)))
    return n.value


def augment(recipe, image, label, scratch=None):
    """Runs one recipe in place on device tensors: image fp32 (channels*D, H, W) or (channels, D, H, W), label fp32 (D, H, W)."""
    import torch
    s = recipe if isinstance(recipe, Recipe) else to_struct(recipe)
    W, H, D = s.dims[0], s.dims[1], s.dims[2]
    if not (image.is_cuda and label.is_cuda):
        raise engine.UNetError("augment: image and label must be device tensors (there is no CPU path)")
    if image.dtype != torch.float32 or label.dtype != torch.float32 or not image.is_contiguous() or not label.is_contiguous():
        raise engine.UNetError("augment: image and label must be contiguous float32")
    if image.numel() != s.channels * D * H * W or label.numel() != D * H * W:
        raise engine.UNetError("augment: tensor sizes do not match the recipe (%d channels of %dx%dx%d)" % (s.channels, W, H, D))
    need = scratch_bytes(s)
    if scratch is None or scratch.numel() * scratch.element_size() < need:
        scratch = torch.empty(need, dtype=torch.uint8, device=image.device)
    st = torch.cuda.current_stream(image.device).cuda_stream
    engine.check(engine.lib.unet_augment_run(C.byref(s), image.data_ptr(), label.data_ptr(), scratch.data_ptr(),
                                              scratch.numel() * scratch.element_size(), st))
    return scratch


def visual_perception_augmentation(options, input, label, is_label, image_shape, random_seed):
    """The reference entry point (visual_perception_augmentation.cpp:163-168): augments `input` (in_count volumes stacked
    along z) and `label` in place.  Both are float32 tensors resident on the GPU."""
    W, H, D = (int(v) for v in image_shape)
    channels = input.numel() // (W * H * D)
    augment(make_recipe(options, image_shape, channels, is_label, random_seed, label_depth=D), input, label)


class AugmentedVolumes:
    """Sample source for `Trainer`: what reader thread B does at train.cpp:439-480 with `source` standing in for the
    template files -- take sample i, augment it with seed = i (in_file_seed), hand {1,in,D,H,W} fp32 + {1,D,H,W} int64."""

    def __init__(self, source, options=None, is_label=True, base_seed=0):
        self.source, self.options, self.is_label, self.base_seed = source, options, is_label, base_seed
        self._scratch = None

    def __call__(self, index):
        import torch
        x, t = self.source(index)
        x = x.clone()
        lab = t[0].to(torch.float32)
        D, H, W = x.shape[2:]
        r = make_recipe(self.options, (W, H, D), x.shape[1], self.is_label, self.base_seed + index, label_depth=D)
        self._scratch = augment(r, x.view(-1), lab.view(-1), self._scratch)
        return x, lab.to(torch.int64)[None]


class PrefetchedVolumes:
    """Device-resident sample ring: the role of the reader / augmentation threads' `file_ready / data_ready` ring
    (train.cpp:259-486: samples are produced ahead of the trainer, which only waits on a flag) on ONE GPU -- sample i + stride is
    produced (template -> augmentation / simulate_modality kernels) on a side stream into buffers of its own while the training
    stream runs step i, and `__call__(i)` hands over the sample prepared earlier after making the training stream wait for its
    event.  `stride` = how far apart this rank's consecutive requests are (world_size when ranks take every world-th sample).
    Results are the synchronous source's bit for bit: the recipe is a function of the sample index alone."""

    def __init__(self, source, stride=1, device=None):
        import torch
        self.source, self.stride = source, int(stride)
        self.stream = torch.cuda.Stream(device=device, priority=0)
        self._ready = {}

    def _produce(self, index):
        import torch
        self.stream.wait_stream(torch.cuda.current_stream(self.stream.device))   # the templates the source reads are final
        with torch.cuda.stream(self.stream):
            x, t = self.source(index)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._ready[index] = (x, t, ev)

    def __call__(self, index):
        import torch
        if index not in self._ready:
            self._produce(index)                     # first request (or a jump): produced now, still off the training stream
        x, t, ev = self._ready.pop(index)
        cur = torch.cuda.current_stream(x.device)
        cur.wait_event(ev)
        x.record_stream(cur)                         # allocated on the side stream, consumed on the training stream
        t.record_stream(cur)
        if len(self._ready) < 2:
            self._produce(index + self.stride)       # runs beside the step that consumes `index`
        return x, t


# ---- simulate_modality (train.cpp:43-178) --------------------------------------------------------------------------------
SIM_TERMS, SIM_MAX_LABELS = 20, 256


class SimRecipe(C.Structure):
    """UnetSimulateRecipe, field for field."""
    _fields_ = [
        ("dims", C.c_int * 3), ("with_label", C.c_int), ("max_label", C.c_int), ("lut", C.c_float * SIM_MAX_LABELS),
        ("term_a", C.c_ubyte * SIM_TERMS), ("term_b", C.c_ubyte * SIM_TERMS), ("term_c", C.c_ubyte * SIM_TERMS),
        ("term_d", C.c_ubyte * SIM_TERMS), ("term_w", C.c_float * SIM_TERMS), ("gamma", C.c_float),
    ]


engine._sig("unet_simulate_modality_scratch_bytes", C.c_int, C.POINTER(SimRecipe), C.POINTER(C.c_size_t))
engine._sig("unet_simulate_modality_run", C.c_int, C.POINTER(SimRecipe), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
EXPORTS += ["unet_simulate_modality_scratch_bytes", "unet_simulate_modality_run"]


def make_simulate_recipe(image_shape, max_label, seed, rand_int=None, rand_float=None):
    """The draws of simulate_modality in the reference's order (train.cpp:52-80): two generators, tipl::uniform_dist<int>(seed)
    and tipl::uniform_dist<float>(0,1,seed+1) -- stand-ins here (numpy MT19937), parity unpinned as for the augmentation.
    max_label = None selects the overload without a label volume (train.cpp:119)."""
    if rand_int is None:
        ri = np.random.RandomState(int(seed) & 0xFFFFFFFF)
        rand_int = lambda n: int(ri.randint(0, n))
    if rand_float is None:
        rf = np.random.RandomState((int(seed) + 1) & 0xFFFFFFFF)
        rand_float = lambda: F(rf.uniform(0.0, 1.0))
    r = {"dims": [int(v) for v in image_shape], "with_label": int(max_label is not None), "max_label": int(max_label or 0)}
    if max_label is not None and not 0 <= max_label < SIM_MAX_LABELS:
        raise ValueError("max_label must be in 0..%d" % (SIM_MAX_LABELS - 1))
    r["lut"] = [F(F(0.4) + rand_float() * F(0.2)) for _ in range(r["max_label"] + 1)] if max_label is not None else []   # :56-58
    r["terms"] = []
    for _ in range(SIM_TERMS):                       # :65-78
        while True:
            a, b = rand_int(4), rand_int(4)
            if a + b != 0:
                break
        c, d = rand_int(4), rand_int(4)
        r["terms"].append((a, b, c, d, rand_float()))
    r["gamma"] = F(F(0.6) + F(1.2) * rand_float())   # :80
    return r


def sim_to_struct(r):
    s = SimRecipe()
    s.dims[:] = r["dims"]
    s.with_label, s.max_label, s.gamma = r["with_label"], r["max_label"], float(r["gamma"])
    for k, v in enumerate(r["lut"]):
        s.lut[k] = float(v)
    for k, (a, b, c, d, w) in enumerate(r["terms"]):
        s.term_a[k], s.term_b[k], s.term_c[k], s.term_d[k], s.term_w[k] = a, b, c, d, float(w)
    return s


def simulate(recipe, t1w, label=None, scratch=None):
    """Runs one recipe in place on device tensors: t1w fp32 (D,H,W) in [0,1]; label fp32 (D,H,W) when the recipe has labels."""
    import torch
    s = recipe if isinstance(recipe, SimRecipe) else sim_to_struct(recipe)
    n = s.dims[0] * s.dims[1] * s.dims[2]
    if not t1w.is_cuda or t1w.dtype != torch.float32 or not t1w.is_contiguous() or t1w.numel() != n:
        raise engine.UNetError("simulate: t1w must be a contiguous float32 device tensor of the recipe's size (there is no CPU path)")
    if s.with_label and (label is None or not label.is_cuda or label.dtype != torch.float32 or not label.is_contiguous() or label.numel() != n):
        raise engine.UNetError("simulate: the recipe needs a float32 device label volume of the same size")
    need = C.c_size_t()
    engine.check(engine.lib.unet_simulate_modality_scratch_bytes(C.byref(s), C.byref(need)))
    if scratch is None or scratch.numel() * scratch.element_size() < need.value:
        scratch = torch.empty(need.value, dtype=torch.uint8, device=t1w.device)
    st = torch.cuda.current_stream(t1w.device).cuda_stream
    engine.check(engine.lib.unet_simulate_modality_run(C.byref(s), t1w.data_ptr(), label.data_ptr() if s.with_label else None,
                                                       scratch.data_ptr(), scratch.numel() * scratch.element_size(), st))
    return scratch


def simulate_modality(t1w, label=None, max_label=None, seed=0):
    """The reference entry points (train.cpp:43-46 / :119): in place on a device volume already normalised to [0,1]."""
    D, H, W = t1w.shape[-3:]
    simulate(make_simulate_recipe((W, H, D), max_label if label is not None else None, seed), t1w, label)
