"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float32 step by step) of the reference's on-GPU augmentation
kernels, visual_perception_augmentation.cu:6-280 and their call sequence .cu:313-521, as a deterministic function of a
recipe (the dict `unet-studio_amd/augment.py:make_recipe` produces; field meaning in include/unet_augment.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this file; the product never does.

PARITY UNPINNED for the parts TIPL defines and the reference tree does not contain (the reference has no test or
fixture for this path either): what is restated from the reference's own text is every kernel body
of the .cu file and the order of its stages; what is OUR definition, shared with the HIP kernels, is
  * tipl::compose_mapping<linear> / tipl::resample / tipl::scale: trilinear, a position is inside when 0 <= p <= dim-1 on
    every axis (NaN is outside), the upper neighbour is clamped to dim-1, outside -> 0; scale maps index*(src/dst);
  * tipl::compose_mapping<majority>: the corner value with the largest summed trilinear weight, first corner on ties;
  * tipl::normalize(I[,upper]): I / max(I) * upper when max > 0; tipl::lower_threshold(I,0): max(I,0);
    tipl::preserve(I,mask): I where mask != 0 else 0; tipl::masking(I,mask): the same on the pre-view image;
  * the noise stream (the reference uses curand XORWOW seeded 0; here a counter hash).
"""
import math

import numpy as np

F = np.float32


def _grid(W, H, D):
    z, y, x = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    return x.astype(F), y.astype(F), z.astype(F)


def _locate(px, py, pz, W, H, D):
    with np.errstate(invalid="ignore"):
        ok = (px >= 0) & (py >= 0) & (pz >= 0) & (px <= F(W - 1)) & (py <= F(H - 1)) & (pz <= F(D - 1))
    px, py, pz = (np.where(ok, p, F(0)).astype(F) for p in (px, py, pz))
    fx, fy, fz = np.floor(px), np.floor(py), np.floor(pz)
    t = (px - fx, py - fy, pz - fz)
    i0 = (fx.astype(np.int64), fy.astype(np.int64), fz.astype(np.int64))
    i1 = (np.minimum(i0[0] + 1, W - 1), np.minimum(i0[1] + 1, H - 1), np.minimum(i0[2] + 1, D - 1))
    return ok, t, i0, i1


def _lerp(t, a, b):
    return a + t * (b - a)


def _trilinear(vol, loc):
    ok, (tx, ty, tz), (x0, y0, z0), (x1, y1, z1) = loc
    at = lambda x, y, z: vol[z, y, x]
    c00 = _lerp(tx, at(x0, y0, z0), at(x1, y0, z0))
    c10 = _lerp(tx, at(x0, y1, z0), at(x1, y1, z0))
    c01 = _lerp(tx, at(x0, y0, z1), at(x1, y0, z1))
    c11 = _lerp(tx, at(x0, y1, z1), at(x1, y1, z1))
    v = _lerp(tz, _lerp(ty, c00, c10), _lerp(ty, c01, c11))
    return np.where(ok, v, F(0)).astype(F)


def _majority(vol, loc):
    ok, (tx, ty, tz), (x0, y0, z0), (x1, y1, z1) = loc
    wx, wy, wz = (F(1) - tx, tx), (F(1) - ty, ty), (F(1) - tz, tz)
    xs, ys, zs = (x0, x1), (y0, y1), (z0, z1)
    v = [vol[zs[i >> 2], ys[(i >> 1) & 1], xs[i & 1]] for i in range(8)]
    w = [wx[i & 1] * wy[(i >> 1) & 1] * wz[i >> 2] for i in range(8)]
    best, best_score = v[0].copy(), np.full(v[0].shape, -1, F)
    for j in range(8):
        s = np.zeros(v[0].shape, F)
        for i in range(8):
            s = s + np.where(v[i] == v[j], w[i], F(0))
        better = s > best_score
        best_score = np.where(better, s, best_score)
        best = np.where(better, v[j], best)
    return np.where(ok, best, F(0)).astype(F)


def _affine(m, x, y, z):
    sr, sh = (np.asarray(m[0], F), np.asarray(m[1], F))
    return (sr[0] * x + sr[1] * y + sr[2] * z + sh[0], sr[3] * x + sr[4] * y + sr[5] * z + sh[1],
            sr[6] * x + sr[7] * y + sr[8] * z + sh[2])


def _hash_u01(index, seed):
    index = index.astype(np.uint64)
    m = np.uint64(0xFFFFFFFF)
    h = (index & m) ^ (((index >> np.uint64(32)) * np.uint64(0x9E3779B9)) & m) ^ np.uint64((seed * 0x85EBCA6B + 0x27D4EB2F) & 0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & m
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & m
    h ^= h >> np.uint64(16)
    return ((h >> np.uint64(8)) + np.uint64(1)).astype(F) * F(1.0 / 16777216.0)


def _scale(src, dims):
    """tipl::scale(src, dst): dst of shape dims (w,h,d)."""
    sd, sh, sw = src.shape
    dw, dh, dd = dims
    x, y, z = _grid(dw, dh, dd)
    px = np.minimum(x * (F(sw) / F(dw)), F(sw - 1))
    py = np.minimum(y * (F(sh) / F(dh)), F(sh - 1))
    pz = np.minimum(z * (F(sd) / F(dd)), F(sd - 1))
    return _trilinear(src, _locate(px, py, pz, sw, sh, sd))


def _norm(v, mx, upper=F(1)):
    return (v / mx * F(upper)).astype(F) if mx > 0 else v


def _perlin_grad(h, x, y, z):   # .cu:204-209
    h = h & 15
    u = np.where(h < 8, x, y)
    v = np.where(h < 4, y, np.where((h == 12) | (h == 14), x, z))
    return np.where(h & 1, -u, u) + np.where(h & 2, -v, v)


def _fade(t):
    return t * t * t * (t * (t * F(6.0) - F(15.0)) + F(10.0))


def _perlin_at(p, x, y, z):   # .cu:211-247
    flx, fly, flz = np.floor(x), np.floor(y), np.floor(z)
    xi, yi, zi = (flx.astype(np.int64) & 255, fly.astype(np.int64) & 255, flz.astype(np.int64) & 255)
    xf, yf, zf = x - flx, y - fly, z - flz
    u, v, w = _fade(xf), _fade(yf), _fade(zf)
    a, b = p[xi] + yi, p[xi + 1] + yi
    aaa, aba, aab, abb = p[p[a] + zi], p[p[a + 1] + zi], p[p[a] + zi + 1], p[p[a + 1] + zi + 1]
    baa, bba, bab, bbb = p[p[b] + zi], p[p[b + 1] + zi], p[p[b] + zi + 1], p[p[b + 1] + zi + 1]
    one = F(1)
    x1 = _lerp(u, _perlin_grad(aaa, xf, yf, zf), _perlin_grad(baa, xf - one, yf, zf))
    x2 = _lerp(u, _perlin_grad(aba, xf, yf - one, zf), _perlin_grad(bba, xf - one, yf - one, zf))
    y1 = _lerp(v, x1, x2)
    x1 = _lerp(u, _perlin_grad(aab, xf, yf, zf - one), _perlin_grad(bab, xf - one, yf, zf - one))
    x2 = _lerp(u, _perlin_grad(abb, xf, yf - one, zf - one), _perlin_grad(bbb, xf - one, yf - one, zf - one))
    y2 = _lerp(v, x1, x2)
    return _lerp(w, y1, y2).astype(F)


def augment(r, image, label):
    """image: float32 (channels, D, H, W); label: float32 (D, H, W).  Returns (image', label') -- what the reference leaves
    in input_ / label_ at .cu:529-530."""
    W, H, D = r["dims"]
    C = r["channels"]
    image = np.array(image, F).reshape(C, D, H, W)
    label = np.array(label, F).reshape(D, H, W)
    maxdim = max(W, H, D)
    x, y, z = _grid(W, H, D)
    N = W * H * D

    if r["downsample"]:   # .cu:315-331
        for c in range(C):
            image[c] = _scale(_scale(image[c], r["low_dims"]), (W, H, D))

    # ---- element-wise stage, .cu:333-380 ----
    cropped = np.zeros((D, H, W), bool)
    if r["crop"]:   # cropping_at_kernel .cu:6-24
        rad = F(r["crop_radius"])
        dx, dy, dz = x - F(r["crop_pos"][0]), y - F(r["crop_pos"][1]), z - F(r["crop_pos"][2])
        ln = np.sqrt(dx * dx + dy * dy + dz * dz)
        cropped = (label != 0) & ~((dx > rad) | (dy > rad) | (dz > rad)) & ~(ln > rad)
    cut = (z < r["trunc_bottom"]) | (z >= D - r["trunc_top"])   # .cu:31-59
    light = spec = None
    if r["diffuse"]:   # diffuse_light_cuda .cu:79-96
        f = np.array(r["diffuse_dir"], F)
        ln = np.sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2])
        if ln != 0:
            f = f / ln
        f = f * (F(r["diffuse_mag"]) / F(maxdim))
        d = (x - F(W) * F(0.5)) * f[0] + (y - F(H) * F(0.5)) * f[1] + (z - F(D) * F(0.5)) * f[2]
        light = np.maximum(F(0), F(1) + d)
    if r["specular"]:  # specular_light_cuda .cu:99-116
        mag = F(r["specular_mag"])
        b = F(1) - mag - mag
        freq = F(float(r["specular_freq"]) * (math.pi * 0.5 / maxdim))
        dx, dy, dz = x - F(r["specular_pos"][0]), y - F(r["specular_pos"][1]), z - F(r["specular_pos"][2])
        ln = np.sqrt(dx * dx + dy * dy + dz * dz)
        spec = (np.cos(ln * freq) + F(1)) * mag + b
    for c in range(C):
        v = image[c]
        if c == 0:   # the channel loop clears the label while cropping channel 0 (.cu:339-340), so later channels find none
            v = np.where(cropped, F(r["crop_value"]), v)
        v = np.where(cut, F(0), v)
        if r["noise"]:   # add_noise_kernel .cu:63-72
            idx = np.arange(N, dtype=np.uint64).reshape(D, H, W) + np.uint64(c * N)
            v = v + F(r["noise_mag"]) * _hash_u01(idx, r["noise_seed"])
        if r["ambient"]:
            v = v + F(r["ambient_value"])
        if light is not None:
            v = v * light
        if spec is not None:
            v = v * spec
        image[c] = v
    label = np.where(cropped | cut, F(0), label).astype(F)

    # ---- view: lens + foci + perspective + affine, .cu:118-189, 383-446 ----
    px, py, pz = x.copy(), y.copy(), z.copy()
    with np.errstate(invalid="ignore", divide="ignore"):
        if r["has_lens"]:
            radius = F(maxdim // 2)
            lm = F(r["lens_magnitude"]) / (radius * radius)
            dx, dy, dz = px - F(W // 2), py - F(H // 2), pz - F(D // 2)
            k = -lm * (dx * dx + dy * dy + dz * dz)
            ax, ay, az = dx * k, dy * k, dz * k
            for f in range(r["n_foci"]):
                rad = F(r["foci_radius"][f])
                r5 = rad * F(r["foci_magnitude"][f])
                pi_r = F(math.pi / float(rad))
                ex, ey, ez = px - F(r["foci_pos"][f][0]), py - F(r["foci_pos"][f][1]), pz - F(r["foci_pos"][f][2])
                ln = np.sqrt(ex * ex + ey * ey + ez * ez)
                inside = ~((ex > rad) | (ey > rad) | (ez > rad)) & ~(ln > rad)
                s = -r5 * np.sin(ln * pi_r) / ln
                ax = np.where(inside, ax + ex * s, ax)
                ay = np.where(inside, ay + ey * s, ay)
                az = np.where(inside, az + ez * s, az)
            px, py, pz = px + ax, py + ay, pz + az
        if r["has_perspective"]:
            p = np.array(r["perspective"], F)
            q = p[0] * (px - F(W) / F(2)) + p[1] * (py - F(H) / F(2)) + p[2] * (pz - F(D) / F(2)) + F(1)
            px, py, pz = px / q, py / q, pz / q
        px, py, pz = _affine(r["view"], px, py, pz)
    loc = _locate(px, py, pz, W, H, D)
    out_label = _majority(label, loc) if r["is_label"] else _trilinear(label, loc)
    out = np.stack([np.fmax(_trilinear(image[c], loc), F(0)) for c in range(C)])
    view_max = [out[c].max() for c in range(C)]

    if r["is_label"] and r["zero_background"]:   # .cu:452-457
        res = np.stack([np.where(out_label != 0, _norm(out[c], view_max[c]), F(0)) for c in range(C)])
        return res.astype(F), out_label
    if not (r["is_label"] and (r["rubber"] or r["perlin"])):
        return np.stack([_norm(out[c], view_max[c]) for c in range(C)]).astype(F), out_label

    # ---- background stage, .cu:460-519 ----
    bg_vox = out_label == 0
    tex = None
    if r["perlin"]:
        p = np.asarray(r["perm"]).astype(np.int64)
        acc, pw = np.zeros((D, H, W), F), F(1)
        for _ in range(4):
            scale = F(r["perlin_zoom"]) * pw
            acc = acc + _perlin_at(p, x * scale, y * scale, z * scale) * pw
            pw = pw * F(0.5)
        acc = acc * F(2)
        acc = acc - np.floor(acc)
        tex = _norm(acc, acc.max(), r["perlin_mag"])
    res = []
    for c in range(C):
        v = _norm(out[c], view_max[c])
        if r["rubber"]:
            masked = np.where(label != 0, image[c], F(0)).astype(F)   # tipl::masking(image,label) .cu:469
            for s in range(5):
                sx, sy, sz = _affine(r["stamp"][s], x, y, z)
                bg = np.fmax(_trilinear(masked, _locate(sx, sy, sz, W, H, D)), F(0))
                bg = _norm(bg, bg.max(), r["stamp_mag"][c][s])
                v = np.where(bg_vox, v + bg * np.maximum(F(0.1), F(1) - v), v)   # blend_kernel .cu:191-198
        if tex is not None:
            v = np.where(bg_vox, v + tex * np.maximum(F(0.1), F(1) - v), v)
        v = np.fmax(v, F(0))
        res.append(_norm(v, v.max()))
    return np.stack(res).astype(F), out_label


def _smooth(v):
    """The smoothing shared with the HIP kernel (k_sim_smooth): 3x3x3 binomial (1,2,1)^3/64, border voxels replicated, taps
    accumulated in (kz, ky, kx) order.  Stands in for tipl::filter::gaussian (TIPL, absent: parity unpinned)."""
    D, H, W = v.shape
    p = np.pad(v, 1, mode="edge")
    acc = np.zeros_like(v)
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                w = F((2 if kz == 1 else 1) * (2 if ky == 1 else 1) * (2 if kx == 1 else 1)) * F(1.0 / 64.0)
                acc = acc + w * p[kz:kz + D, ky:ky + H, kx:kx + W]
    return acc.astype(F)


def simulate_modality(r, t1w, label=None):
    """train.cpp:43-117 (with labels) / :119-178 (without), as a function of the recipe of unet-studio_amd/augment.py."""
    t1w = np.array(t1w, F)
    if r["with_label"]:
        lut = np.array(r["lut"], F)
        tissue = lut[np.clip(label.astype(np.int64), 0, r["max_label"])]          # :59-60
    else:
        tissue = t1w.copy()                                                        # :126
    tissue = _smooth(_smooth(tissue))                                              # :62-63
    x, z = t1w, tissue
    rx, rz = F(1) - x, F(1) - z
    one = np.ones_like(x)
    px, pz = [one, x, x * x, x * x * x], [one, z, z * z, z * z * z]
    qx, qz = [one, rx, rx * rx, rx * rx * rx], [one, rz, rz * rz, rz * rz * rz]
    s = np.zeros_like(x)
    for a, b, c, d, w in r["terms"]:
        s = s + F(w) * px[a] * pz[b] * qx[c] * qz[d]
    keep = ~(x <= F(0.02))
    with np.errstate(invalid="ignore"):
        out = np.where(keep, np.power(s, F(r["gamma"])), F(0)).astype(F)          # :84-104
    sel = keep & ((label != 0) if r["with_label"] else True)
    if sel.any():
        mn, mx = out[sel].min(), out[sel].max()
        if mx > mn:                                                                # :110-115
            out = np.clip((out - mn) * (F(1) / (mx - mn)), F(0), F(1)).astype(F)
    return out
