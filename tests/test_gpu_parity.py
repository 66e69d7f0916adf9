"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against the oracle and the golden vectors.
Tolerances: fp32 engine 1e-4 relative (north_star: logits within 1e-4 rel of the CPU reference);
bf16 engine: its own measured error, bounded here at 4e-2 relative (8 mantissa bits through ~12 layers)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import unet_studio_amd as U  # noqa: E402
from oracle import aten_ref as A  # noqa: E402
from oracle import oracle as O  # noqa: E402

E = U.engine
DEV = "cuda:0"
TOL = {"fp32": 1e-4, "bf16": 4e-2}
TDT = {"fp32": torch.float32, "bf16": torch.bfloat16}
EDT = {"fp32": U.DTYPE_F32, "bf16": U.DTYPE_BF16}


def rel(a, b, scale=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = float(np.abs(b).max()) if scale is None else scale
    return float(np.abs(a - b).max()) / max(scale, 1e-30)


def stream():
    return torch.cuda.current_stream(DEV).cuda_stream


def to_cl(x, dt):  # numpy [C,D,H,W] -> device channels-last [D,H,W,C]
    return torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 2, 3, 0)))).to(DEV).to(TDT[dt]).contiguous()


def from_cl(t):  # device [D,H,W,C] -> numpy [C,D,H,W] fp32
    return np.ascontiguousarray(np.transpose(t.float().cpu().numpy(), (3, 0, 1, 2)))


def scratch(cin, cout, D=1, H=1, W=1):
    b = C.c_size_t()
    E.check(E.lib.unet_op_scratch_bytes(cin, cout, D, H, W, C.byref(b)))
    return torch.empty(b.value, dtype=torch.uint8, device=DEV)


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def q(x, dt):  # what the engine sees after rounding its input to the element type
    return x if dt == "fp32" else torch.from_numpy(x).to(torch.bfloat16).float().numpy()


CONV_CASES = [  # cin, cout, (D,H,W), ks, stride
    (1, 16, (9, 10, 12), 3, 1), (16, 16, (8, 8, 16), 3, 1), (5, 7, (6, 7, 9), 3, 1), (16, 32, (8, 10, 12), 3, 2),
    (3, 4, (7, 9, 11), 3, 2), (16, 6, (5, 6, 7), 1, 1), (32, 16, (4, 8, 16), 3, 1), (24, 40, (5, 5, 6), 3, 1),
    # MFMA-eligible shapes: every tile configuration (W >= 12, 5..11, <= 4), both channel-chunk widths, NT 1/2/4, ragged edges
    (32, 32, (9, 7, 20), 3, 1), (64, 64, (6, 9, 8), 3, 1), (48, 16, (5, 6, 4), 3, 1), (16, 48, (12, 12, 13), 3, 1),
    (128, 64, (4, 4, 4), 3, 1), (16, 16, (3, 5, 7), 3, 1), (32, 16, (16, 16, 32), 3, 1),
    # stride-2 MFMA wgrad: every tile configuration (Wo >= 12, 5..11, <= 4), PJ 1/2/4, odd input sizes
    (16, 32, (12, 16, 32), 3, 2), (32, 64, (9, 11, 13), 3, 2), (64, 16, (8, 8, 8), 3, 2), (16, 16, (5, 7, 25), 3, 2),
    # output-stationary wgrad that adds straight into the gradient (tile side 4^3 / 8^3: no slab, no reduce): stride 2 with an 8^3
    # output from an even and from an odd input (4 tiles along z per block), stride 1 at 8^3 with several (ca, cb) pairs
    (32, 32, (16, 16, 16), 3, 2), (16, 32, (15, 15, 15), 3, 2), (48, 32, (8, 8, 8), 3, 1),
    # stride-2 forward with 32-channel chunks (outputs of 8^3 voxels or fewer): eight and four chunks, both tiles, several row tiles
    (256, 64, (8, 8, 8), 3, 2), (128, 48, (16, 16, 16), 3, 2), (96, 32, (13, 12, 15), 3, 2),
    # register-accumulating small wgrads: first conv (Cin = 1; W % 4 != 0 falls back to the row kernel) and 1x1x1 heads
    (1, 8, (6, 5, 8), 3, 1), (1, 16, (5, 6, 7), 3, 1), (1, 32, (17, 9, 16), 3, 1), (64, 6, (4, 5, 6), 1, 1), (256, 6, (3, 4, 5), 1, 1),
    (32, 3, (5, 6, 7), 1, 1), (16, 8, (33, 8, 9), 1, 1),
    # small-volume MFMA kernel: 8 chunks (two per wave, fragment refill), 16 chunks (two super-stages), 4^3 tile
    (256, 16, (8, 8, 8), 3, 1), (512, 32, (5, 6, 7), 3, 1), (256, 32, (4, 4, 4), 3, 1), (96, 32, (3, 4, 10), 3, 1),
    # sliding-window MFMA wgrad (bf16, stride 1, W >= 24): every (ca-tiles, cb-tiles) pairing 1x1 / 2x1 / 1x2 / 2x2, several pair
    # groups per launch, ragged footprints in y and x, z segments of unequal length, volumes narrower than one 32-voxel row
    (16, 16, (9, 11, 37), 3, 1), (32, 16, (13, 9, 33), 3, 1), (16, 32, (5, 12, 40), 3, 1), (32, 32, (6, 9, 40), 3, 1),
    (64, 32, (8, 8, 24), 3, 1), (48, 16, (4, 10, 70), 3, 1), (16, 16, (40, 8, 32), 3, 1), (32, 64, (11, 5, 29), 3, 1),
    # ... above 32^3 voxels (at or below, the kernel uses single pairs): the 2x2, 2x1 and 1x2 pair blocks
    (32, 32, (20, 40, 48), 3, 1), (32, 16, (24, 30, 48), 3, 1), (16, 32, (18, 40, 48), 3, 1),
    # sliding window for a single 16-channel chunk (bf16; forward Cin = 16, dgrad Cout = 16; W >= 12, D >= 8): one and two row tiles,
    # ragged footprints, z segments of unequal length
    (16, 16, (11, 9, 19), 3, 1), (16, 32, (10, 17, 33), 3, 1), (32, 16, (9, 10, 18), 3, 1), (16, 16, (37, 8, 16), 3, 1),
    # stride-2 sliding-window kernels (kernels_mfma_s2.hip; bf16, output >= 16 wide, >= 4 deep): forward with 16- and 32-channel planes and
    # 1 / 2 / 4 row tiles per block, dgrad with 32 and 64 dy channels and 1..3 row-tile blocks, odd input sizes (last fine plane / row /
    # voxel without a partner), ragged footprints, z segments of unequal length; Cin 48 / Cout 16 fall back to the halo-tile kernel
    (16, 32, (16, 20, 40), 3, 2), (32, 64, (9, 13, 35), 3, 2), (16, 16, (8, 8, 32), 3, 2), (32, 48, (11, 10, 33), 3, 2),
    (16, 64, (12, 9, 38), 3, 2), (48, 32, (8, 12, 32), 3, 2), (32, 32, (37, 7, 31), 3, 2),
    # ... and their sliding-window weight gradient (kernels_mfma_s2_wgrad.hip; output >= 24 wide): (ca, cb) pairings 1x2 / 2x2 / 2x1, several
    # pair groups, ragged 32-voxel K-steps, odd input sizes, z segments of unequal length
    (16, 32, (9, 12, 64), 3, 2), (32, 32, (12, 9, 50), 3, 2), (32, 16, (8, 8, 48), 3, 2), (16, 64, (21, 7, 70), 3, 2),
    # split-K kernels of the deep levels (kernels_mfma_deep.hip; bf16, Cin % 32 == 0): 27-tap kinds at <= 64 output voxels (ragged
    # last 64-voxel group, one- and two-voxel extents, K splits of 8..32, one and two row tiles per block), stride 2 from odd extents,
    # and their dgrads (stride-2 dgrad up to a 512-voxel coarse grid)
    (64, 48, (3, 4, 5), 3, 1), (128, 32, (2, 2, 2), 3, 1), (512, 64, (1, 2, 3), 3, 1), (32, 128, (4, 4, 3), 3, 1),
    (32, 64, (7, 6, 5), 3, 2), (128, 128, (8, 8, 8), 3, 2), (64, 32, (3, 2, 2), 3, 2), (32, 32, (13, 15, 16), 3, 2),
    # fp32 matrix-core conv (fp32 engine, volumes >= 4096 voxels): NT 1 / 2, ragged tile edges in z, y and x, 8-channel chunk tail
    (16, 16, (16, 16, 32), 3, 1), (32, 64, (17, 19, 21), 3, 1), (24, 48, (9, 23, 22), 3, 1), (64, 32, (18, 17, 16), 3, 1),
]


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("impl", [U.IMPL_DIRECT, U.IMPL_AUTO])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_ops(case, dt, impl):
    cin, cout, (D, H, W), ks, st = case
    l = O.lib()
    x = q(rnd((cin, D, H, W), 1), dt); w = rnd((cout, cin, ks, ks, ks), 2, 0.2); b = rnd((cout,), 3)
    pad = (ks - 1) // 2
    od = [(s + 2 * pad - ks) // st + 1 for s in (D, H, W)]
    y_ref = np.empty((cout, *od), np.float32)
    l.orc_conv3d_fwd(O._f(x), O._f(w), O._f(b), O._f(y_ref), cin, cout, D, H, W, ks, st)
    sc = scratch(cin, cout, D, H, W)
    wd, bd = torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV)
    xd = to_cl(x, dt)
    yd = torch.empty((*od, cout), dtype=TDT[dt], device=DEV)
    E.check(E.lib.unet_op_conv3d_fwd(EDT[dt], impl, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), cin, cout, D, H, W,
                                     ks, st, sc.data_ptr(), stream()))
    assert rel(from_cl(yd), y_ref) < (1e-5 if dt == "fp32" else 1e-2)
    # dgrad / wgrad
    dy = q(rnd((cout, *od), 4), dt)
    dx_ref = np.empty_like(x); dw_ref = np.zeros_like(w); db_ref = np.zeros_like(b)
    l.orc_conv3d_bwd_data(O._f(dy), O._f(w), O._f(dx_ref), cin, cout, D, H, W, ks, st)
    l.orc_conv3d_bwd_weight(O._f(x), O._f(dy), O._f(dw_ref), O._f(db_ref), cin, cout, D, H, W, ks, st)
    dyd = to_cl(dy, dt)
    dxd = torch.empty((D, H, W, cin), dtype=TDT[dt], device=DEV)
    E.check(E.lib.unet_op_conv3d_bwd_data(EDT[dt], impl, dyd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), cin, cout, D, H, W, ks, st,
                                          sc.data_ptr(), stream()))
    assert rel(from_cl(dxd), dx_ref) < (1e-5 if dt == "fp32" else 1e-2)
    dwd = torch.ones_like(wd); dbd = torch.ones_like(bd)  # += semantics: start from 1
    E.check(E.lib.unet_op_conv3d_bwd_weight(EDT[dt], impl, xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), dbd.data_ptr(), cin, cout,
                                            D, H, W, ks, st, sc.data_ptr(), stream()))
    assert rel(dwd.cpu().numpy() - 1.0, dw_ref) < (2e-5 if dt == "fp32" else 1e-2)
    assert rel(dbd.cpu().numpy() - 1.0, db_ref) < (2e-5 if dt == "fp32" else 1e-2)


@pytest.mark.parametrize("case", [(1, 16, (9, 13, 21)), (1, 32, (4, 8, 16)), (32, 16, (6, 9, 20)), (16, 16, (5, 8, 17)),
                                  (16, 16, (9, 13, 21)), (16, 32, (12, 9, 20)), (16, 16, (33, 8, 16))])
def test_conv3d_plain_with_statistics(case):
    """plain input (what a plan feeds its convs: activated copies), bf16, statistics epilogue: the first-conv MFMA kernel
    (Cin = 1) and the persistent kernel"""
    cin, cout, (D, H, W) = case
    l = O.lib()
    x = q(rnd((cin, D, H, W), 1), "bf16"); w = rnd((cout, cin, 3, 3, 3), 2, 0.2); b = rnd((cout,), 3)
    y_ref = np.empty((cout, D, H, W), np.float32)
    l.orc_conv3d_fwd(O._f(x), O._f(w), O._f(b), O._f(y_ref), cin, cout, D, H, W, 3, 1)
    sc = scratch(cin, cout, D, H, W)
    wd, bd, xd = torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV), to_cl(x, "bf16")
    yd = torch.empty((D, H, W, cout), dtype=torch.bfloat16, device=DEV)
    stats = torch.empty((cout, 2), dtype=torch.float32, device=DEV)
    E.check(E.lib.unet_op_conv3d_fwd_fused(U.DTYPE_BF16, U.IMPL_AUTO, xd.data_ptr(), None, None, 0, wd.data_ptr(), bd.data_ptr(),
                                           yd.data_ptr(), stats.data_ptr(), cin, cout, D, H, W, 3, 1, sc.data_ptr(), stream()))
    y = from_cl(yd)
    assert rel(y, y_ref) < 1e-2
    st = stats.cpu().numpy()
    assert np.allclose(st[:, 0], y.reshape(cout, -1).sum(1), rtol=1e-4, atol=1e-2)
    assert np.allclose(st[:, 1], (y.reshape(cout, -1) ** 2).sum(1), rtol=1e-4, atol=1e-2)


def test_conv3d_pack_then_kernel_only():
    """unet_op_conv3d_pack + unet_op_conv3d_fwd_packed (what bench.py times as the dominant kernel) = unet_op_conv3d_fwd;
    shapes the MFMA kernels do not cover are refused"""
    cin, cout, D, H, W = 32, 16, 6, 9, 20
    l = O.lib()
    x = q(rnd((cin, D, H, W), 1), "bf16"); w = rnd((cout, cin, 3, 3, 3), 2, 0.2); b = rnd((cout,), 3)
    y_ref = np.empty((cout, D, H, W), np.float32)
    l.orc_conv3d_fwd(O._f(x), O._f(w), O._f(b), O._f(y_ref), cin, cout, D, H, W, 3, 1)
    wp, part = scratch(cin, cout, D, H, W), scratch(cin, cout, D, H, W)
    wd, bd, xd = torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV), to_cl(x, "bf16")
    yd = torch.empty((D, H, W, cout), dtype=torch.bfloat16, device=DEV)
    E.check(E.lib.unet_op_conv3d_pack(U.DTYPE_BF16, wd.data_ptr(), wp.data_ptr(), cin, cout, D, H, W, 3, 1, stream()))
    E.check(E.lib.unet_op_conv3d_fwd_packed(U.DTYPE_BF16, xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), yd.data_ptr(), part.data_ptr(),
                                            cin, cout, D, H, W, 3, 1, stream()))
    assert rel(from_cl(yd), y_ref) < 1e-2
    assert E.lib.unet_op_conv3d_pack(U.DTYPE_F32, wd.data_ptr(), wp.data_ptr(), cin, cout, D, H, W, 3, 1, stream()) != 0
    assert b"MFMA" in E.lib.unet_last_error()


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("impl", [U.IMPL_DIRECT, U.IMPL_AUTO])
@pytest.mark.parametrize("case", [(16, 16, (6, 9, 20), 3, 1, 2), (32, 32, (8, 8, 8), 3, 2, 3), (16, 32, (5, 6, 4), 3, 1, 1), (5, 7, (4, 5, 6), 3, 1, 2)])
def test_conv3d_fused_prologue_epilogue(case, dt, impl):
    """norm + activation applied by the consumer while it reads (zero padding AFTER the activation), and the
    {sum, sum of squares} of the stored output that the next norm needs (unet.cpp:74-98 fused into unet.cpp:59-72)"""
    cin, cout, (D, H, W), ks, st, act = case
    l = O.lib()
    x = q(rnd((cin, D, H, W), 1), dt); w = rnd((cout, cin, ks, ks, ks), 2, 0.2); b = rnd((cout,), 3)
    scale = (1.0 + 0.3 * rnd((cin,), 5)).astype(np.float32); shift = rnd((cin,), 6, 0.5)
    xn = x * scale[:, None, None, None] + shift[:, None, None, None]
    xa = np.empty_like(xn)
    l.orc_act_fwd(O._f(np.ascontiguousarray(xn)), O._f(xa), C.c_int64(xa.size), act)
    xa = q(xa, dt)   # the engine rounds the transformed value to the element type before the MFMA
    od = [(s + 2 - ks) // st + 1 for s in (D, H, W)]
    y_ref = np.empty((cout, *od), np.float32)
    l.orc_conv3d_fwd(O._f(xa), O._f(w), O._f(b), O._f(y_ref), cin, cout, D, H, W, ks, st)
    sc = scratch(cin, cout, D, H, W)
    wd, bd, xd = torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV), to_cl(x, dt)
    sd, hd = torch.from_numpy(scale).to(DEV), torch.from_numpy(shift).to(DEV)
    yd = torch.empty((*od, cout), dtype=TDT[dt], device=DEV)
    stats = torch.empty((cout, 2), dtype=torch.float32, device=DEV)
    E.check(E.lib.unet_op_conv3d_fwd_fused(EDT[dt], impl, xd.data_ptr(), sd.data_ptr(), hd.data_ptr(), act, wd.data_ptr(), bd.data_ptr(),
                                           yd.data_ptr(), stats.data_ptr(), cin, cout, D, H, W, ks, st, sc.data_ptr(), stream()))
    y = from_cl(yd)
    assert rel(y, y_ref) < (1e-5 if dt == "fp32" else 1.5e-2)
    got = stats.cpu().numpy().astype(np.float64)
    yy = y.astype(np.float64).reshape(cout, -1)
    assert np.allclose(got[:, 0], yy.sum(1), rtol=1e-4, atol=1e-3 * np.abs(yy).sum(1).max())
    assert np.allclose(got[:, 1], (yy * yy).sum(1), rtol=1e-4)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(16, 16, (4, 5, 6)), (7, 3, (3, 4, 5)), (32, 16, (4, 4, 8)),
                                  # MFMA conv_trans (Cin % 32 == 0): every tile configuration, ragged edges
                                  (64, 32, (3, 5, 19)), (32, 32, (4, 4, 4)), (128, 64, (2, 3, 2)), (32, 48, (5, 9, 7)),
                                  # output-stationary conv_trans wgrad (coarse side 4^3 above, 8^3 here: 4 tiles along z per block)
                                  (32, 48, (8, 8, 8)),
                                  # conv_trans dgrad with 32-channel chunks (Cout % 32 == 0 on coarse grids of 8^3 or less): both tiles
                                  (64, 64, (5, 6, 7)), (32, 96, (8, 8, 8)), (256, 256, (4, 4, 4)),
                                  # split-K kernels of the deep levels (coarse grids of <= 512 voxels, Cin % 32 == 0): ragged groups, one-voxel extents
                                  (64, 32, (3, 3, 5)), (32, 16, (1, 2, 3)), (256, 128, (2, 2, 2)), (96, 64, (7, 8, 8)),
                                  # sliding-window conv_trans kernels (kernels_mfma_s2.hip; coarse grid >= 16 wide, >= 4 deep): forward with 1 / 2 / 4
                                  # k-steps and 1..4 row-tile blocks, dgrad with 16- and 32-channel planes and 2 / 4 row tiles per block, ragged edges
                                  (32, 16, (4, 6, 16)), (64, 32, (5, 9, 19)), (128, 64, (4, 4, 16)), (32, 48, (6, 7, 17)), (64, 16, (7, 5, 20)),
                                  (32, 32, (9, 4, 18)),
                                  # ... and their sliding-window weight gradient (coarse grid >= 24 wide): pairings 1x2 / 2x2, several pair groups, ragged
                                  (32, 16, (4, 5, 24)), (64, 32, (5, 3, 33)), (32, 32, (4, 4, 26)), (32, 48, (9, 6, 25))])
def test_convt_ops(case, dt):
    cin, cout, (D, H, W) = case
    l = O.lib()
    x = q(rnd((cin, D, H, W), 1), dt); w = rnd((cin, cout, 2, 2, 2), 2, 0.3); b = rnd((cout,), 3)
    y_ref = np.empty((cout, 2 * D, 2 * H, 2 * W), np.float32)
    l.orc_convt_fwd(O._f(x), O._f(w), O._f(b), O._f(y_ref), cin, cout, D, H, W)
    sc = scratch(cin, cout, D, H, W)
    wd, bd, xd = torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV), to_cl(x, dt)
    yd = torch.empty((2 * D, 2 * H, 2 * W, cout), dtype=TDT[dt], device=DEV)
    E.check(E.lib.unet_op_convt_fwd(EDT[dt], 0, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), cin, cout, D, H, W,
                                    sc.data_ptr(), stream()))
    assert rel(from_cl(yd), y_ref) < (1e-5 if dt == "fp32" else 1e-2)
    dy = q(rnd(y_ref.shape, 4), dt)
    dx_ref = np.empty_like(x); dw_ref = np.zeros_like(w); db_ref = np.zeros_like(b)
    l.orc_convt_bwd_data(O._f(dy), O._f(w), O._f(dx_ref), cin, cout, D, H, W)
    l.orc_convt_bwd_weight(O._f(x), O._f(dy), O._f(dw_ref), O._f(db_ref), cin, cout, D, H, W)
    dyd = to_cl(dy, dt)
    dxd = torch.empty((D, H, W, cin), dtype=TDT[dt], device=DEV)
    E.check(E.lib.unet_op_convt_bwd_data(EDT[dt], 0, dyd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), cin, cout, D, H, W,
                                         sc.data_ptr(), stream()))
    assert rel(from_cl(dxd), dx_ref) < (1e-5 if dt == "fp32" else 1e-2)
    dwd = torch.ones_like(wd); dbd = torch.ones_like(bd)   # += semantics: start from 1
    E.check(E.lib.unet_op_convt_bwd_weight(EDT[dt], 0, xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), dbd.data_ptr(), cin, cout,
                                           D, H, W, sc.data_ptr(), stream()))
    assert rel(dwd.cpu().numpy() - 1.0, dw_ref) < (2e-5 if dt == "fp32" else 1e-2)
    assert rel(dbd.cpu().numpy() - 1.0, db_ref) < (2e-5 if dt == "fp32" else 1e-2)


def load_case(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    n = len([k for k in d.files if k.startswith("param") and not k.startswith("param_")])
    return d, n


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["cfg1_bnorm_16", "mix_16", "c16_24"])
def test_network_against_golden(golden_dir, name, dt):
    """forward logits, fused losses, every parameter gradient, running stats, one optimizer step, both eval modes"""
    d, n = load_case(golden_dir, name)
    arch, cin, cout = str(d["arch"]), int(d["cin"]), int(d["cout"])
    tol = TOL[dt]
    m = U.UNet3d(cin, cout, arch, device=DEV, dtype=dt)
    m.load_parameters([d["param%d" % i] for i in range(n)])
    x = torch.from_numpy(d["x"])[None].to(DEV)
    t = torch.from_numpy(d["target"])[None].to(DEV)
    m.train()
    plan = m.plan_for(x.shape[2:]); ws = m._workspace(plan)
    outs = m._run_forward(plan, ws, x, 1)
    k = 0
    while "logits%d" % k in d.files:
        assert rel(outs[k][0].cpu().numpy(), d["logits%d" % k]) < tol, "logits level %d" % k
        k += 1
    losses, gouts = m.loss(outs, t)
    lo = losses.cpu().numpy()
    assert abs(lo[0] - float(d["loss"])) < tol * max(1.0, float(d["loss"]))
    assert np.allclose(lo[1:], d["stats"], rtol=tol, atol=tol)
    m._run_backward(plan, ws, gouts)
    gmax = max(float(np.abs(d["grad%d" % i]).max()) for i in range(n))
    gt = 2e-4 if dt == "fp32" else 8e-2
    for i, g in enumerate(m.grads()):
        assert rel(g.cpu().numpy(), d["grad%d" % i], gmax) < gt, "grad %d" % i
    for i, b in enumerate(m.buffers()):  # running_mean / running_var (num_batches_tracked is host-side)
        ref = [d["buffer_after%d" % j] for j in range(len([f for f in d.files if f.startswith("buffer_after")])) if d["buffer_after%d" % j].dtype.kind == "f"][i]
        assert np.allclose(b.cpu().numpy(), ref, rtol=10 * tol, atol=tol), "buffer %d" % i
    if dt == "fp32":
        opt = m.create_optimizer(float(d["lr"]))
        opt.step(grad_scale=1.0 / int(d["batch_size"]))
        assert abs(float(opt.last_grad_norm) - float(d["grad_norm"])) < 2e-4 * float(d["grad_norm"])
        for i, p in enumerate(m.parameters()):
            ref = d["param_after%d" % i]
            assert rel(p.cpu().numpy(), ref, max(1e-3, float(np.abs(ref).max()))) < 1e-5, "param %d" % i
        assert float(m.flat_grads.abs().max()) == 0.0  # zero_grad
    # eval modes on the post-step fixture parameters
    nbuf = [d[k] for k in sorted((f for f in d.files if f.startswith("buffer_after")), key=lambda s: int(s[12:])) if d[k].dtype.kind == "f"]
    m.load_parameters([d["param_after%d" % i] for i in range(n)], nbuf)
    m.eval()
    with torch.no_grad():
        assert rel(m.forward(x)[0][0].cpu().numpy(), d["eval_logits0"]) < tol
        m.prepare_for_inference()
        assert rel(m.forward(x)[0][0].cpu().numpy(), d["infer_logits0"]) < tol


GRAD_SAMPLE_STRIDE = 997   # = tests/golden/make_golden.py


def _check_grad_samples(m, d, big, tol, min_numel=0):
    off, worst = 0, 0.0
    for i, g in enumerate(m.grads()):
        smp = g.flatten()[::GRAD_SAMPLE_STRIDE].cpu().numpy()
        want = d["grad_sample"][off:off + smp.size]
        off += smp.size
        if not big[i] or g.numel() < min_numel:
            continue
        err = float(np.abs(smp - want).max()) / float(d["grad_absmax"][i])
        worst = max(worst, err)
        assert err < tol, "gradient sample of parameter %d: %.3e of the tensor's max" % (i, err)
    assert off == d["grad_sample"].size
    return worst


def _check_grad_samples_l2(m, d, big, tol, tol_el, min_numel=0, min_samples=32):
    """the strided sample of every gradient tensor as a VECTOR: relative L2 distance to the fixture's sample (a permuted tap or a
    transposed channel pair gives ~1.4; rounding noise averages over the sample's elements) where the sample holds >= min_samples elements,
    else every sampled element against tol_el of the tensor's max -> (worst L2 distance, worst single element over all tensors)"""
    off, worst, worst_el, rows = 0, 0.0, 0.0, []
    for i, g in enumerate(m.grads()):
        smp = g.flatten()[::GRAD_SAMPLE_STRIDE].cpu().numpy().astype(np.float64)
        want = d["grad_sample"][off:off + smp.size].astype(np.float64)
        off += smp.size
        if not big[i] or g.numel() < min_numel:
            continue
        err = float(np.linalg.norm(smp - want) / max(np.linalg.norm(want), 1e-30))
        el = float(np.abs(smp - want).max()) / float(d["grad_absmax"][i])
        rows.append((i, smp.size, err, el))
        worst_el = max(worst_el, el)
        if smp.size >= min_samples:
            worst = max(worst, err)
    assert off == d["grad_sample"].size
    print("  gradient samples (parameter, sampled elements, relative L2 of the sample, worst element / tensor max):",
          [(i, n, "%.2e" % e, "%.2e" % x) for i, n, e, x in sorted(rows, key=lambda r: -r[2])[:6]])
    for i, n, e, x in rows:
        if n >= min_samples:
            assert e < tol, "gradient sample vector of parameter %d (%d sampled elements): relative L2 distance %.3e" % (i, n, e)
        else:
            assert x < tol_el, "gradient sample of parameter %d (%d sampled elements): %.3e of the tensor's max" % (i, n, x)
    return worst, worst_el


def _logit_stride(n, level):   # = tests/golden/make_golden.py:logit_stride
    side, st = n >> level, 1
    while side // st > 16:
        st *= 2
    return st if n > 64 else (4 if level == 0 else 1)


@pytest.mark.parametrize("size", [64, 128])
def test_default_arch_fp32_against_golden(golden_dir, size):
    """default architecture (train.cpp:1054-1069) at 64^3 and at BASELINE.json's 128^3 (config 2, the fp32 parity
    configuration: logits within 1e-4 relative of the CPU reference), weights = ATen module init under manual_seed(0)"""
    d = np.load(os.path.join(golden_dir, "default_arch_%d.npz" % size))
    n = int(d["n"])
    torch.manual_seed(0)
    ref = A.UNet3dRef(1, 6, A.default_feature(6))
    pl2 = np.array([float(p.detach().double().norm()) for p in ref.parameters()])
    if not np.allclose(pl2, d["param_l2"], rtol=1e-6):
        pytest.fail("module init under manual_seed(0) differs from the fixture's (different torch build?)")
    m = U.UNet3d(1, 6, A.default_feature(6), device=DEV, dtype="fp32")
    m.load_parameters([p.detach().numpy() for p in ref.parameters()])
    x, t = A.synthetic_sample(1, 6, (n, n, n), 1)
    x, t = x.to(DEV), t.to(DEV)
    plan = m.plan_for(x.shape[2:]); ws = m._workspace(plan)
    outs = m._run_forward(plan, ws, x, 1)
    for k in range(5):
        a = outs[k][0].cpu().numpy()
        st = _logit_stride(n, k)
        assert rel(a[:, ::st, ::st, ::st], d["logits%d" % k]) < 1e-4, "logits level %d" % k
        assert abs(np.sqrt((a.astype(np.float64) ** 2).sum()) - float(d["logits_l2_%d" % k])) < 1e-4 * float(d["logits_l2_%d" % k])
    losses, gouts = m.loss(outs, t)
    assert abs(float(losses[0]) - float(d["loss"])) < 1e-4 * float(d["loss"])
    m._run_backward(plan, ws, gouts)
    gl2 = np.array([float(g.double().norm()) for g in m.grads()])
    big = d["grad_l2"] > 1e-3 * d["grad_l2"].max()   # conv biases before a norm have analytically zero gradient
    # 2e-3 relative, plus 2e-4 of the largest norm for the small ones: a leaky_relu voxel on its kink takes another slope when the
    # forward convs sum in another order (the fp32 matrix-core kernel) -- see test_noncubic_network_against_live_aten, whose elu twin
    # shows the engine itself at 5e-7 of fp64 gradients
    assert np.allclose(gl2[big], d["grad_l2"][big], rtol=2e-3, atol=2e-4 * float(d["grad_l2"].max()))
    heads = np.stack([np.pad(g.flatten()[:16].cpu().numpy(), (0, max(0, 16 - g.numel()))) for g in m.grads()])
    assert rel(heads[big], d["grad_head"][big]) < 2e-3
    # every GRAD_SAMPLE_STRIDE-th element of every parameter gradient, each tensor against its own largest magnitude: catches a
    # gradient that is wrong but keeps its norm (a permuted tap, a transposed channel pair), which the norms and leading elements miss.
    # Asserted at BASELINE's 128^3 only: at 64^3 the two deepest levels normalise over 8 and 64 voxels, where one leaky_relu voxel on
    # its kink moves single gradient elements by percents (measured 3.2e-3, 5.2e-3 and 2.6e-2 of the tensor's max on parameters 3, 62
    # and 66) -- there the worst tensor is printed, not bounded.
    worst = _check_grad_samples(m, d, big, 1e-2 if size == 128 else 1.0)
    print("fp32 gradient samples at %d^3: worst tensor %.3e of its max" % (size, worst))


def test_default_arch_128_bf16_against_golden(golden_dir):
    """the benchmarked configuration itself (bf16 activations, fp32 master weights, default arch, 128^3) against the CPU
    reference's fixture: logits, loss and the large gradients within the bf16 engine's measured error (bounds as in DESIGN.md 5)"""
    d = np.load(os.path.join(golden_dir, "default_arch_128.npz"))
    n = int(d["n"])
    torch.manual_seed(0)
    ref = A.UNet3dRef(1, 6, A.default_feature(6))
    m = U.UNet3d(1, 6, A.default_feature(6), device=DEV, dtype="bf16")
    m.load_parameters([p.detach().numpy() for p in ref.parameters()])
    x, t = A.synthetic_sample(1, 6, (n, n, n), 1)
    x, t = x.to(DEV), t.to(DEV)
    plan = m.plan_for(x.shape[2:]); ws = m._workspace(plan)
    outs = m._run_forward(plan, ws, x, 1)
    for k in range(5):
        a = outs[k][0].cpu().numpy()
        st = _logit_stride(n, k)
        # bf16 rounding noise grows with depth: the 8^3 / 4^3 levels normalise over 512 / 64 voxels only (measured 4.8e-2 at level 4)
        assert rel(a[:, ::st, ::st, ::st], d["logits%d" % k]) < (4e-2 if k < 3 else 8e-2), "logits level %d" % k
    losses, gouts = m.loss(outs, t)
    assert abs(float(losses[0]) - float(d["loss"])) < 1e-2 * float(d["loss"])
    m._run_backward(plan, ws, gouts)
    gl2 = np.array([float(g.double().norm()) for g in m.grads()])
    big = d["grad_l2"] > 1e-2 * d["grad_l2"].max()
    # norms of the filter gradients within 8 %; the 16..256-element norm parameters' (sums over up to 2M voxels of cancelling bf16 terms)
    # within 15 %: between two valid kernel sets (sliding-window / halo-tile / split-K, profiles/r20_deep_vs_sliding_gradients.txt, r20_sliding_vs_halo_gradients.txt) such a
    # norm moves by up to 5 % and this fixture's encode1.1.bias sits 4..9 % from the fp32 reference depending on the set
    numel = np.array([g.numel() for g in m.grads()])
    assert np.allclose(gl2[big & (numel >= 1024)], d["grad_l2"][big & (numel >= 1024)], rtol=8e-2)
    assert np.allclose(gl2[big], d["grad_l2"][big], rtol=0.15)
    # bf16 engine: its measured element-wise error on the conv / conv_trans weights (>= 1024 elements).  The 16..256-element norm and
    # bias gradients are sums over up to 2M voxels of bf16-rounded, cancelling terms: one sampled element of such a tensor was 25 % of
    # the tensor's max away (their norms are bounded above); a permuted or transposed filter gradient is off by O(1) on every sample.
    # (bound 0.4: measured 0.24 on encode1.0.weight, whose dL/dy has crossed the whole network in bf16; the fp32 engine passes the same
    # samples at 9e-3 and the per-layer operator tests hold every element of a bf16 weight gradient to 1e-2 of the oracle's)
    worst = _check_grad_samples(m, d, big, 0.4, min_numel=1024)
    print("bf16 gradient samples: worst tensor %.3e of its max" % worst)


@pytest.mark.parametrize("size", [64, 128])
def test_default_arch_bf16_against_the_bf16_storage_oracle(golden_dir, size):
    """The benchmarked configuration (bf16 activations / gradients, fp32 master weights, default architecture; 128^3 = BASELINE configs[2])
    against the oracle WITH ROUNDING HOOKS: the same ATen fp32 kernels, rounding to bf16 exactly where the engine stores bf16
    (oracle/aten_ref.py:run_bf16_storage; fixture default_arch_<n>_bf16.npz).  What is left between the two is summation order
    inside a layer and the roundings it flips, so the bounds are several times tighter than against the fp32 reference
    (test_default_arch_128_bf16_against_golden: 4e-2 / 8e-2 logits, 0.4 of a tensor's max on gradient samples): logits 1e-2 at the two
    finest levels and 2e-2 below, loss 2e-3, filter-gradient norms 3e-2 (measured 1.4e-2), sampled elements of every gradient tensor of >= 1024
    elements as a sample vector, relative L2 0.3 (measured 0.15-0.23), and the small tensors (norm scales / shifts, biases: < 1024 elements) by their norm, 0.15."""
    d = np.load(os.path.join(golden_dir, "default_arch_%d_bf16.npz" % size))
    n = int(d["n"])
    torch.manual_seed(0)
    ref = A.UNet3dRef(1, 6, A.default_feature(6))
    m = U.UNet3d(1, 6, A.default_feature(6), device=DEV, dtype="bf16")
    m.load_parameters([p.detach().numpy() for p in ref.parameters()])
    x, t = A.synthetic_sample(1, 6, (n, n, n), 1)
    x, t = x.to(DEV), t.to(DEV)
    plan = m.plan_for(x.shape[2:]); ws = m._workspace(plan)
    outs = m._run_forward(plan, ws, x, 1)
    worst_logit = 0.0
    for k in range(5):
        a = outs[k][0].cpu().numpy()
        st = _logit_stride(n, k)
        e = rel(a[:, ::st, ::st, ::st], d["logits%d" % k])
        worst_logit = max(worst_logit, e)
        print("  level %d logits: %.3e of the level's largest" % (k, e))
        # measured: 2e-3 .. 6e-3 at the 128^3 .. 32^3 levels, 1.0e-2 .. 1.4e-2 at the 16^3 / 8^3 levels (their norms average over 512 / 64
        # voxels at the bottom of the network: one flipped rounding there moves a whole channel)
        assert e < (1e-2 if k < 2 else 2e-2), "logits level %d: %.3e" % (k, e)
        assert abs(np.sqrt((a.astype(np.float64) ** 2).sum()) - float(d["logits_l2_%d" % k])) < 2e-3 * float(d["logits_l2_%d" % k])
    losses, gouts = m.loss(outs, t)
    assert abs(float(losses[0]) - float(d["loss"])) < 2e-3 * float(d["loss"])
    m._run_backward(plan, ws, gouts)
    gl2 = np.array([float(g.double().norm()) for g in m.grads()])
    # which tensors carry a gradient at all: a conv bias in front of a norm has an analytically zero gradient -- the fp32 fixture says
    # which (in bf16 both sides hold rounding noise there, of the size of a small real gradient)
    d32 = np.load(os.path.join(golden_dir, "default_arch_%d.npz" % size))
    big = (d["grad_l2"] > 1e-2 * d["grad_l2"].max()) & (d32["grad_l2"] > 1e-3 * d32["grad_l2"].max())
    numel = np.array([g.numel() for g in m.grads()])
    rel_l2 = np.abs(gl2 - d["grad_l2"]) / np.maximum(d["grad_l2"], 1e-30)
    large, small = big & (numel >= 1024), big & (numel < 1024)
    order = np.argsort(-np.where(large, rel_l2, 0))[:3]
    order_s = np.argsort(-np.where(small, rel_l2, 0))[:3]
    print("  worst gradient norms (parameter, elements, relative error): filters", [(int(i), int(numel[i]), "%.3e" % rel_l2[i]) for i in order],
          " small tensors", [(int(i), int(numel[i]), "%.3e" % rel_l2[i]) for i in order_s])
    assert rel_l2[large].max() < 3e-2, "gradient norm of parameter %d: %.3e" % (int(order[0]), rel_l2[large].max())
    # small tensors (norm scales / shifts, biases of 16..256 elements): by their norm -- a sampled element of a 16-element tensor says
    # little.  Each is a sum over up to 2M voxels of cancelling bf16 terms; both sides carry ~0.5 % of rounding-flip noise on every term
    # by the time the gradient has crossed the network, and the sums cancel to about a tenth of their terms' random-walk size (measured:
    # 8.8e-2 on encode0.4.weight at 128^3, 7.7e-2 on encode1.4.bias at 64^3).  A wrong statistic (mean / rstd / a dropped term) is off by O(1).
    assert rel_l2[small].max() < 0.15, "gradient norm of parameter %d: %.3e" % (int(order_s[0]), rel_l2[small].max())
    # the strided sample of every filter gradient (>= 1024 elements; every 997th element) as a vector: relative L2 distance to the
    # oracle's sample.  Measured 0.15 .. 0.23 on EVERY filter tensor, whether the sample holds 7 or 1775 elements -- i.e. it does not
    # average out: with the synthetic sample's random labels a filter-gradient element is itself a random-walk sum over up to 2M voxels,
    # and the ~0.5 % of rounding-flip noise each bf16 term carries moves every element by ~1/5 of its size, while the tensors' NORMS
    # (asserted above) agree to 1.4e-2 and the per-kernel operator tests hold every element of every weight gradient to 1e-2.  Bound 0.3:
    # a permuted tap or a transposed channel pair moves the sample vector by ~1.4.  Worst single elements: 7e-2 .. 1.4e-1 of the
    # tensor's max (the bound against the fp32 reference is 0.4).
    worst, worst_el = _check_grad_samples_l2(m, d, big, 0.3, 0.35, min_numel=1024, min_samples=1)
    print("bf16 vs bf16-storage oracle at %d^3: logits %.3e, gradient norms %.3e, gradient sample vectors %.3e (relative L2), worst single "
          "sampled element %.3e of its tensor's max" % (size, worst_logit, rel_l2[big].max(), worst, worst_el))


def test_default_arch_64_fp32_elu_twin_gradient_samples(golden_dir):
    """The fp32 engine's gradients on the default architecture at 64^3 with every leaky_relu replaced by elu (fixture
    default_arch_64_elu.npz): with a smooth activation no voxel sits on a kink, so sampled gradient ELEMENTS are bounded (1e-4 of each
    tensor's max) where test_default_arch_fp32_against_golden[64] can only print them (its 8^3 / 4^3 levels normalise over 512 / 64
    voxels and one leaky_relu voxel on its kink moves single elements by percents)."""
    d = np.load(os.path.join(golden_dir, "default_arch_64_elu.npz"))
    n = int(d["n"])
    arch = A.default_feature(6).replace("leaky_relu", "elu")
    torch.manual_seed(0)
    ref = A.UNet3dRef(1, 6, arch)
    m = U.UNet3d(1, 6, arch, device=DEV, dtype="fp32")
    m.load_parameters([p.detach().numpy() for p in ref.parameters()])
    x, t = A.synthetic_sample(1, 6, (n, n, n), 1)
    x, t = x.to(DEV), t.to(DEV)
    plan = m.plan_for(x.shape[2:]); ws = m._workspace(plan)
    outs = m._run_forward(plan, ws, x, 1)
    for k in range(5):
        a = outs[k][0].cpu().numpy()
        st = _logit_stride(n, k)
        assert rel(a[:, ::st, ::st, ::st], d["logits%d" % k]) < 1e-4, "logits level %d" % k
    losses, gouts = m.loss(outs, t)
    assert abs(float(losses[0]) - float(d["loss"])) < 1e-4 * float(d["loss"])
    m._run_backward(plan, ws, gouts)
    gl2 = np.array([float(g.double().norm()) for g in m.grads()])
    big = d["grad_l2"] > 1e-3 * d["grad_l2"].max()
    assert np.allclose(gl2[big], d["grad_l2"][big], rtol=1e-4, atol=1e-5 * float(d["grad_l2"].max()))
    worst = _check_grad_samples(m, d, big, 1e-4)
    print("fp32 elu twin at 64^3: worst gradient sample %.3e of its tensor's max" % worst)


ARCH_NONCUBIC = ("conv16,ks3,stride1+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu\n"
                 "conv32,ks3,stride2+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu\n"
                 "conv64,ks3,stride2+norm,leaky_relu+conv64,ks3,stride1+norm,leaky_relu+conv_trans32,ks2,stride2\n"
                 "conv32,ks3,stride1+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu+conv5,ks1,stride1+conv_trans16,ks2,stride2\n"
                 "conv16,ks3,stride1+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu+conv5,ks1,stride1")


ARCH_MIX_NC = ("conv8,ks3,stride1+norm,elu+conv8,ks3,stride1+norm,leaky_relu\n"
               "conv16,ks3,stride2+norm,elu+conv16,ks3,stride1+norm,leaky_relu\n"
               "max_pool+conv16,ks3,stride1+norm,relu+upsample\n"
               "conv16,ks3,stride1+norm,leaky_relu+conv5,ks1,stride1+conv_trans8,ks2,stride2\n"
               "conv8,ks3,stride1+norm,leaky_relu+conv5,ks1,stride1")


ARCH_DEEP2 = ("conv32,ks3,stride1+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu\n"
              "conv64,ks3,stride2+norm,leaky_relu+conv64,ks3,stride1+norm,leaky_relu+conv_trans32,ks2,stride2\n"
              "conv32,ks3,stride1+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu+conv5,ks1,stride1")


@pytest.mark.parametrize("norm", ["norm", "bnorm"])
@pytest.mark.parametrize("size", [(8, 8, 8), (12, 8, 10), (16, 16, 16)])
def test_deep_level_kernels_network_against_live_aten(size, norm):
    """kernels_mfma_deep.hip against the ATen CPU executor run live in float64: a two-level network of 32 / 64 channels whose coarse
    level is 4^3 (every kind of the path incl. both norm epilogues), 6x4x5 (120 voxels: a ragged last 64-voxel group, odd extents under
    the stride-2 conv and the conv_trans) or 8^3 (the short kinds only), with InstanceNorm3d and with BatchNorm3d (eps 0, running
    statistics).  bf16 against fp64: the bounds of test_noncubic_network_against_live_aten's bf16 row, gradients a little wider (64-voxel
    norms: measured values are printed)."""
    arch = ARCH_DEEP2.replace("norm", norm)
    torch.manual_seed(5)
    ref = A.UNet3dRef(2, 5, arch)
    ref.train()
    x, t = A.synthetic_sample(2, 5, size, 13)
    m = U.UNet3d(2, 5, arch, device=DEV, dtype="bf16")
    m.load_parameters([p.detach().numpy() for p in ref.parameters()])
    ref = ref.double()
    outs_ref = ref(x.double())
    loss_ref, _ = A.deep_supervision_loss(outs_ref, t, 5)
    loss_ref.backward()
    xd, td = x.to(DEV), t.to(DEV)
    plan = m.plan_for(xd.shape[2:]); ws = m._workspace(plan)
    outs = m._run_forward(plan, ws, xd, 1)
    errs = []
    for k, (o, r) in enumerate(zip(outs, outs_ref)):
        e = rel(o[0].cpu().numpy().astype(np.float64), r[0].detach().numpy())
        errs.append(e)
        assert e < 8e-2, "logits level %d: %g" % (k, e)
    losses, gouts = m.loss(outs, td)
    assert abs(float(losses[0]) - float(loss_ref)) < 2e-2 * float(loss_ref)
    m._run_backward(plan, ws, gouts)
    gref = torch.cat([p.grad.flatten() for p in ref.parameters()]).numpy()
    g = m.flat_grads.cpu().numpy().astype(np.float64)
    e = np.abs(g - gref).max() / np.abs(gref).max()
    print("deep-level network %s %s: logits %s, gradients %.6g of the largest" % (size, norm, ", ".join("%.6g" % v for v in errs), e))
    assert e < 0.25, "gradients: %g" % e
    if norm == "bnorm":     # the running statistics the training forward moved (momentum 0.1, unbiased variance)
        rbufs = [r for r in ref.buffers() if r.dtype.is_floating_point]     # running_mean / running_var (num_batches_tracked is host-side)
        assert len(rbufs) == len(m.buffers()) > 0
        for i, (b, r) in enumerate(zip(m.buffers(), rbufs)):
            assert rel(b.float().cpu().numpy(), r.detach().float().numpy()) < 3e-2, "buffer %d" % i


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("size", [(24, 40, 56), (20, 36, 12), (8, 16, 132), (12, 20, 28, "mix")])
def test_noncubic_network_against_live_aten(size, dt):
    """volumes that are not cubes and not multiples of any tile (ragged sliding-window columns, partial planes, W < 16, long rows):
    forward, loss and backward against the ATen CPU executor run live IN FLOAT64 (oracle/aten_ref.py).

    Gradients and the kink of leaky_relu / relu: a voxel whose normalised value lies within rounding distance of 0 takes slope 1 in
    one fp32 evaluation and 0.01 in another.  One such voxel moves the weight gradients around it by ~1e-3 of the largest gradient
    (profiles/dbg_grad_error.py: 7e-3 at (20,36,12), 3e-3 at (16,16,36), nothing at (16,16,40); the same digits with every kernel
    variant, with fp32 or fp64 statistics).  With the smooth activation (elu) in the same places the fp32 engine is within 5e-7 of the
    fp64 gradients on every size -- so the fp32 engine is checked twice: the architecture as given with a kink-tolerant bound, and
    its elu twin at fp32 resolution."""
    arch = ARCH_NONCUBIC
    if len(size) == 4:   # every other layer kind (elu / relu, max_pool, upsample, 8-channel convs on the direct kernels)
        arch, size = ARCH_MIX_NC, size[:3]
    variants = [(arch, 1e-2 if dt == "fp32" else 1.5e-1)]
    if dt == "fp32":
        variants.append((arch.replace("leaky_relu", "elu").replace(",relu", ",elu"), 5e-6))
    for arch_v, grad_bound in variants:
        torch.manual_seed(3)
        ref = A.UNet3dRef(2, 5, arch_v)
        ref.train()
        x, t = A.synthetic_sample(2, 5, size, 11)
        m = U.UNet3d(2, 5, arch_v, device=DEV, dtype=dt)
        m.load_parameters([p.detach().numpy() for p in ref.parameters()])
        ref = ref.double()
        outs_ref = ref(x.double())
        loss_ref, _ = A.deep_supervision_loss(outs_ref, t, 5)
        loss_ref.backward()
        xd, td = x.to(DEV), t.to(DEV)
        plan = m.plan_for(xd.shape[2:]); ws = m._workspace(plan)
        outs = m._run_forward(plan, ws, xd, 1)
        for k, (o, r) in enumerate(zip(outs, outs_ref)):
            e = rel(o[0].cpu().numpy().astype(np.float64), r[0].detach().numpy())
            assert e < (1e-4 if dt == "fp32" else 8e-2), "logits level %d: %g" % (k, e)
        losses, gouts = m.loss(outs, td)
        assert abs(float(losses[0]) - float(loss_ref)) < (1e-4 if dt == "fp32" else 2e-2) * float(loss_ref)
        m._run_backward(plan, ws, gouts)
        gref = torch.cat([p.grad.flatten() for p in ref.parameters()]).numpy()
        g = m.flat_grads.cpu().numpy().astype(np.float64)
        e = np.abs(g - gref).max() / np.abs(gref).max()
        assert e < grad_bound, "gradients (%s): %g" % ("as given" if arch_v is arch else "elu twin", e)


def test_backward_in_buckets_equals_one_backward():
    """unet_backward_part over unet_plan_backward_buckets = unet_backward, bit for bit; every bucket callback sees final gradients
    for its element range (what the data-parallel trainer all-reduces under the rest of the backward)"""
    n = 32
    m = U.UNet3d(1, 6, U.default_feature(6), device=DEV, dtype="bf16", seed=0)
    x, t = U.SyntheticVolumes(1, 6, (n, n, n), DEV)(0)
    m.train()
    l1 = m.forward_backward(x, t).clone(); g1 = m.flat_grads.clone()
    m.flat_grads.zero_()
    seen = []

    def on_bucket(lo, hi):
        torch.cuda.synchronize()
        assert torch.equal(m.flat_grads[lo:hi], g1[lo:hi]), "bucket [%d, %d) not final when announced" % (lo, hi)
        seen.append((lo, hi))
    l2 = m.forward_backward_bucketed(x, t, on_bucket)
    assert torch.equal(l1, l2) and torch.equal(m.flat_grads, g1)
    assert len(seen) == 3 and seen[0][1] == m.flat_grads.numel() and seen[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(seen[:-1], seen[1:]))          # contiguous, from the end of the buffer to its start
    assert seen[0][0] == 7600464 and seen[1][0] == 879696                  # decoder | encoder levels 5-4 | encoder levels 3-0
    # an architecture whose parameters are few: one bucket
    m2 = U.UNet3d(1, 3, "conv4\nconv4\nconv3,ks1", device=DEV, dtype="fp32", seed=0)
    assert m2.plan_for((8, 8, 8)).backward_buckets(1) == [(0, 0)]


@pytest.mark.parametrize("collapse", [0, 2])
def test_fused_loss_vs_oracle(collapse):
    """calc_losses (train.cpp:501-552): collapse_before, labels >= out_count masked, each cost switch"""
    arch = "conv4\nconv4\nconv5,ks1"
    m = U.UNet3d(1, 5, arch, device=DEV, dtype="fp32")
    g = torch.Generator().manual_seed(3)
    n = 12
    logits = (3.0 * torch.randn((1, 5, n, n, n), generator=g)).to(DEV)
    target = torch.randint(0, 7, (1, n, n, n), generator=g).to(DEV)
    plan = m.plan_for((n, n, n))
    for ce, dice, mse in ((1, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)):
        losses, gouts = m.loss([logits], target, bool(ce), bool(dice), bool(mse), collapse, plan=plan)
        w = (ce, dice, mse) if (ce or dice or mse) else (1, 0, 0)
        (oce, odice, omse), dl = O.calc_losses(logits[0].cpu().numpy(), target[0].cpu().numpy(), 5, collapse, w, True)
        lo = losses.cpu().numpy()
        assert np.allclose(lo[1:], [oce, odice, omse], rtol=2e-5, atol=2e-5)
        assert abs(lo[0] - (w[0] * oce + w[1] * odice + w[2] * omse)) < 2e-5 * 3
        assert rel(gouts[0][0].cpu().numpy(), dl) < 1e-4


def test_autograd_bridge_matches_fused_path(golden_dir):
    """torch losses can drive backward through forward()'s autograd edge (total_loss.backward(), train.cpp:706)"""
    d, n = load_case(golden_dir, "mix_16")
    m = U.UNet3d(int(d["cin"]), int(d["cout"]), str(d["arch"]), device=DEV, dtype="fp32")
    m.load_parameters([d["param%d" % i] for i in range(n)])
    x = torch.from_numpy(d["x"])[None].to(DEV); t = torch.from_numpy(d["target"])[None].to(DEV)
    m.train()
    outs = m.forward(x)
    loss, _ = A.deep_supervision_loss(outs, t, int(d["cout"]))
    loss.backward()
    gmax = max(float(np.abs(d["grad%d" % i]).max()) for i in range(n))
    for i, g in enumerate(m.grads()):
        assert rel(g.cpu().numpy(), d["grad%d" % i], gmax) < 2e-4
    # gradients accumulate across micro-steps (train.cpp:604-606,706)
    before = m.flat_grads.clone()
    m.forward_backward(x, t)
    assert rel(m.flat_grads.cpu().numpy(), 2 * before.cpu().numpy()) < 1e-4


def test_errors_surface_like_the_reference():
    m = U.UNet3d(1, 2, "conv4\nconv4\nconv2,ks1", device=DEV, dtype="fp32")
    with pytest.raises(U.UNetError):
        m.forward(torch.zeros((1, 3, 8, 8, 8), device=DEV))
    with pytest.raises(U.UNetError, match="invalid collapse_before"):
        m.loss([torch.zeros((1, 2, 8, 8, 8), device=DEV)], torch.zeros((1, 8, 8, 8), dtype=torch.int64, device=DEV), collapse_before=2)


@pytest.mark.parametrize("dt", ["bf16"])
def test_full_size_128_properties(dt):
    """BASELINE.json's size (default arch, 128^3): properties that need no CPU reference.
    (1) bit-exact run-to-run determinism (no float atomics anywhere);
    (2) instance norm makes the net invariant to the scale of a conv that feeds a norm;
    (3) the deep-supervision gradient is consistent with a finite difference of the loss along a random direction."""
    n = 128
    m = U.UNet3d(1, 6, U.default_feature(6), device=DEV, dtype=dt, seed=0)
    x, t = U.SyntheticVolumes(1, 6, (n, n, n), DEV)(0)
    m.train()
    l1 = m.forward_backward(x, t).clone(); g1 = m.flat_grads.clone()
    m.flat_grads.zero_()
    l2 = m.forward_backward(x, t).clone(); g2 = m.flat_grads.clone()
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    with torch.no_grad():
        base = m.forward(x)[0].clone()
        m.parameters()[0].mul_(2.0); m.parameters()[1].mul_(2.0)   # encode0.0 conv feeds a norm
        scaled = m.forward(x)[0]
        assert rel(scaled.cpu().numpy(), base.cpu().numpy()) < 5e-2
        m.parameters()[0].mul_(0.5); m.parameters()[1].mul_(0.5)
    # directional derivative along the gradient itself (fp32 master weights, bf16 activations: loose tolerance; a random
    # direction moves every weight by less than a bf16 ulp and measures rounding noise, not the derivative)
    dvec = g1 / g1.norm()
    eps = 2e-2
    with torch.no_grad():
        m.flat_params.add_(eps * dvec); lp = float(m.loss(m.forward(x), t, want_grad=False)[0][0])
        m.flat_params.add_(-2 * eps * dvec); lm = float(m.loss(m.forward(x), t, want_grad=False)[0][0])
        m.flat_params.add_(eps * dvec)
    fd = (lp - lm) / (2 * eps)
    an = float((g1 * dvec).sum())
    assert abs(fd - an) < 0.25 * max(abs(an), 1e-3), (fd, an)


@pytest.mark.timeout(900)
def test_configs4_network_256_two_channels_augmented_bf16_step():
    """BASELINE configs[4] on one GPU: default architecture, in = 2 (T1+T2), 256^3, bf16, every sample augmented on the device
    (AugmentedVolumes -> unet_augment_run) and one Trainer.step().  No CPU reference finishes at this size, so the checks are
    size-independent properties: (1) the whole optimizer step -- augmentation, forward, loss, backward, clip, SGD -- is bit
    reproducible from the same state; (2) gradients and updated parameters are finite and non-zero; (3) the gradient agrees
    with a central finite difference of the loss along itself (as test_full_size_128_properties)."""
    n = 256
    src = U.SyntheticVolumes(2, 6, (n, n, n), DEV, cache=1)
    feed = U.AugmentedVolumes(lambda i: src(0))

    def one_step():
        m = U.UNet3d(2, 6, U.default_feature(6), device=DEV, dtype="bf16", seed=0)
        tr = U.Trainer(m, U.TrainingParam(batch_size=1, epoch=100, learning_rate=0.01), feed)
        stats = tr.step().clone()
        torch.cuda.synchronize()
        return m, stats, m.flat_params.clone(), float(m.optimizer.last_grad_norm)

    m1, s1, p1, gn1 = one_step()
    init = U.UNet3d(2, 6, U.default_feature(6), device=DEV, dtype="bf16", seed=0).flat_params.clone()
    del m1
    m2, s2, p2, gn2 = one_step()
    assert torch.equal(s1, s2) and torch.equal(p1, p2) and gn1 == gn2, "the 256^3 augmented step is not bit reproducible"
    assert torch.isfinite(p1).all() and torch.isfinite(s1).all() and np.isfinite(gn1) and gn1 > 0
    assert float((p1 - init).abs().max()) > 0, "the step did not move the parameters"
    assert 0.5 < float(s1[0]) < 20.0                      # 5-level CE + Dice + MSE of an untrained 6-class net
    # gradient vs finite difference on the augmented sample (m2 now holds the post-step weights; any point works)
    x, t = feed(0)
    assert x.shape == (1, 2, n, n, n) and t.shape == (1, n, n, n) and int(t.max()) < 6
    m2.flat_grads.zero_()
    m2.forward_backward(x, t)
    g = m2.flat_grads.clone()
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    dvec = g / g.norm()
    eps = 2e-2
    with torch.no_grad():
        m2.flat_params.add_(eps * dvec); m2._params_version += 1
        lp = float(m2.loss(m2.forward(x), t, want_grad=False)[0][0])
        m2.flat_params.add_(-2 * eps * dvec); m2._params_version += 1
        lm = float(m2.loss(m2.forward(x), t, want_grad=False)[0][0])
    fd, an = (lp - lm) / (2 * eps), float((g * dvec).sum())
    assert abs(fd - an) < 0.25 * max(abs(an), 1e-3), (fd, an)


# ---- the two forward-only callers of the path: evaluate.cpp:211-246 (a22) and the validation thread train.cpp:826-852 (a23) ----
ARCH_BN_EVAL = ("conv8,ks3,stride1+bnorm,relu+conv8,ks3,stride1+bnorm,relu\n"
                "max_pool+conv16,ks3,stride1+bnorm,relu+conv16,ks3,stride1+bnorm,relu+conv_trans8,ks2,stride2\n"
                "conv8,ks3,stride1+bnorm,relu+conv8,ks3,stride1+bnorm,relu+conv6,ks1,stride1")


def _trained_bn_pair(dt):
    """engine model + ATen module with the same weights and the same non-trivial running statistics (two train forwards)."""
    torch.manual_seed(5)
    ref = A.UNet3dRef(2, 6, ARCH_BN_EVAL)
    ref.train()
    m = U.UNet3d(2, 6, ARCH_BN_EVAL, device=DEV, dtype=dt)
    m.load_parameters([p.detach().numpy() for p in ref.parameters()])
    m.train()
    for s in range(2):
        x, _ = A.synthetic_sample(2, 6, (16, 24, 32), 20 + s)
        with torch.no_grad():
            ref(x)
        m.forward(x.to(DEV))
    return m, ref


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_validation_forward_keeps_running_statistics(dt):
    """train.cpp:834-852: eval() without prepare_for_inference -> bnorm uses the running statistics; mean (ce, dice, mse) of
    calc_losses on output [0] over the test volumes, appended to testing_errors."""
    m, ref = _trained_bn_pair(dt)
    vols = [A.synthetic_sample(2, 6, (16, 24, 32), 40 + i) for i in range(2)]
    ref.eval()
    exp = np.zeros(3)
    with torch.no_grad():
        for x, t in vols:
            ce, dice, mse = A.calc_losses(ref(x)[0], t, 6)
            exp += np.array([float(ce), float(dice), float(mse)])
    exp /= len(vols)
    tr = U.Trainer(m, U.TrainingParam(batch_size=1, epoch=10), lambda i: None)
    got = tr.validate([x.to(DEV) for x, _ in vols], [t.to(DEV) for _, t in vols])
    tol = 2e-4 if dt == "fp32" else 3e-2
    assert np.allclose(got, exp, rtol=tol, atol=tol * 0.1), (got, exp)
    assert m.get_testing_errors() == got and m._training      # recorded; training mode restored for thread C


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_evaluate_loop_matches_the_reference_inference(dt):
    """evaluate.cpp:386-399,211-246: prepare_for_inference (running statistics reset: bnorm = gamma*x+beta), forward [0],
    host buffer (in*D,H,W) -> (out*D,H,W)."""
    m, ref = _trained_bn_pair(dt)
    ios = [[A.synthetic_sample(2, 6, (16, 24, 32), 60 + i)[0][0].reshape(2 * 16, 24, 32).numpy()] for i in range(2)]
    ios[1].append(A.synthetic_sample(2, 6, (8, 16, 40), 70)[0][0].reshape(2 * 8, 16, 40).numpy())   # a second size in one file
    ev = U.EvaluateUNet(m)
    out = ev.start(ios)
    assert not ev.aborted and ev.error_msg == "" and ev.cur_prog == 2
    ref.prepare_for_inference()
    tol = 2e-4 if dt == "fp32" else 4e-2
    for a, b in zip(ios, out):
        for io, res in zip(a, b):
            d = io.shape[0] // 2
            with torch.no_grad():
                e = ref(torch.from_numpy(io).view(1, 2, d, io.shape[1], io.shape[2]))[0][0].numpy()
            assert res.shape == (6 * d, io.shape[1], io.shape[2]) and res.dtype == np.float32
            assert rel(torch.from_numpy(res.reshape(e.shape)), torch.from_numpy(e)) < tol
    # an input the network cannot take ends the run the reference's way: message + aborted, no exception (evaluate.cpp:234-242)
    bad = ev.start([[np.zeros((2 * 5, 24, 32), np.float32)]])
    assert ev.aborted and ev.error_msg.startswith("error during evaluation:") and ev.cur_prog == 0


def test_nz_file_round_trip_through_the_engine(tmp_path):
    """save_to_file / load_from_file (main.cpp:157-233) around a real model: the reloaded network gives the same logits, and the
    checkpoint written every 100 steps (train.cpp:780-788) carries the error history."""
    arch = ("conv8,ks3,stride1+norm,leaky_relu\nconv16,ks3,stride2+norm,leaky_relu+conv_trans8,ks2,stride2\n"
            "conv8,ks3,stride1+norm,leaky_relu+conv3,ks1,stride1")
    m = U.UNet3d(1, 3, arch, device=DEV, dtype="fp32", seed=3)
    m.training_errors, m.testing_errors = [0.3, 0.2, 0.1], [0.4, 0.3, 0.2]
    m.voxel_size, m.dim = (0.5, 0.5, 2.0), (64, 96, 32)
    f = str(tmp_path / "model.nz")
    assert U.save_to_file(m, f)
    r = U.load_from_file(f, device=DEV, dtype="fp32")
    assert r.architecture == m.architecture and r.dim == m.dim and r.testing_errors == pytest.approx(m.testing_errors)
    x, _ = U.SyntheticVolumes(1, 3, (16, 16, 16), DEV)(0)
    m.eval(); r.eval()
    assert torch.equal(m.forward(x)[0], r.forward(x)[0])


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_fused_forward_loss_equals_the_two_calls(dt):
    """unet_forward_loss (forward + calc_losses in one call, coarse levels' loss on the plan's side stream) against unet_forward
    followed by unet_loss: identical logits and identical dL/dlogits at every level; the total only differs by the order in which
    the five weighted level losses are added."""
    arch = U.default_feature(4)
    m = U.UNet3d(1, 4, arch, device=DEV, dtype=dt, seed=1)
    x, t = U.SyntheticVolumes(1, 4, (32, 32, 32), DEV)(3)
    plan = m.plan_for((32, 32, 32))
    ws = m._workspace(plan)
    outs_a = m._run_forward(plan, ws, x, mode=1)
    losses_a, g_a = m.loss(outs_a, t, True, True, True, 0, plan=plan)
    outs_a = [o.clone() for o in outs_a]; g_a = [g.clone() for g in g_a]; losses_a = losses_a.clone()
    outs_b, losses_b, g_b = m._run_forward_loss(plan, ws, x, t, True, True, True, 0)
    torch.cuda.synchronize()
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    for a, b in zip(g_a, g_b):
        assert torch.equal(a, b)
    assert torch.equal(losses_a[1:], losses_b[1:])                       # level-0 ce, dice, mse
    assert abs(float(losses_a[0]) - float(losses_b[0])) <= 2e-6 * abs(float(losses_a[0]))
    # and it is reproducible from run to run (two streams, fixed summation order)
    _, losses_c, g_c = m._run_forward_loss(plan, ws, x, t, True, True, True, 0)
    torch.cuda.synchronize()
    assert torch.equal(losses_b, losses_c) and all(torch.equal(a, b) for a, b in zip(g_b, g_c))


_BNSTATS_CHILD = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import unet_studio_amd as U
n = int(sys.argv[3])
arch = ("conv16,ks3,stride1+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu\n"
        "conv32,ks3,stride2+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu+conv_trans16,ks2,stride2\n"
        "conv16,ks3,stride1+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu+conv6,ks1,stride1")
if len(sys.argv) > 4 and sys.argv[4] == "default":
    arch = U.default_feature(6)
if len(sys.argv) > 4 and sys.argv[4] == "deep2":      # two levels of 32 / 64 channels: at 16^3 the coarse level is 8^3 (kernels_mfma_deep.hip)
    arch = ("conv32,ks3,stride1+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu\n"
            "conv64,ks3,stride2+norm,leaky_relu+conv64,ks3,stride1+norm,leaky_relu+conv_trans32,ks2,stride2\n"
            "conv32,ks3,stride1+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu+conv6,ks1,stride1")
m = U.UNet3d(1, 6, arch, device="cuda:0", dtype=sys.argv[5] if len(sys.argv) > 5 else "bf16", seed=0)
x, t = U.SyntheticVolumes(1, 6, (n, n, n), "cuda:0", cache=2)(0)
m.forward_backward(x, t)
torch.cuda.synchronize()
np.save(sys.argv[2], m.flat_grads.cpu().numpy())
plan = m.plan_for((n, n, n))
json.dump([[nm, int(np.prod(s))] for nm, s in zip(plan.param_names, plan.param_shapes)], open(sys.argv[2] + ".json", "w"))
"""


def _grads_in_fresh_process(tmp_path, tag, n, env_extra, arch="", dtype="bf16"):
    """flat gradients of one forward + backward of a small bf16 network, computed in a fresh process (the engine reads its
    experiment switches once per process) -> (gradients, [(parameter name, element count)])"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / ("grads_%s.npy" % tag))
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", _BNSTATS_CHILD, root, path, str(n), arch, dtype], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    return np.load(path), json.load(open(path + ".json"))


def _by_name(g, names):
    off, seen = 0, {}
    for nm, cnt in names:
        seen[nm] = g[off:off + cnt]
        off += cnt
    return seen


def test_norm_backward_statistics_in_the_dgrad_epilogue_match_the_separate_pass(tmp_path):
    """The matrix-core dgrads leave the norm-backward statistics of a single-consumer norm layer in the epilogue of the dgrad that
    produces its gradient (engine.cpp: BnBwdStats); UNET_NO_DGRAD_BNSTATS=1 keeps k_norm_bwd_stats8 as a separate pass.  The switch is
    read once per process, so both variants run in fresh processes.  The first fused layer of the backward (decode0.1: everything
    before it is bit-identical) must agree to summation-order noise; later layers see bf16 roundings of du flip, which is why the
    rest is only bounded loosely."""
    g1, names = _grads_in_fresh_process(tmp_path, "fused", 16, {})
    g0, _ = _grads_in_fresh_process(tmp_path, "separate", 16, {"UNET_NO_DGRAD_BNSTATS": "1"})
    a1, a0 = _by_name(g1, names), _by_name(g0, names)
    for nm in ("decode0.1.weight", "decode0.1.bias"):
        assert np.abs(a1[nm] - a0[nm]).max() <= 2e-6 * np.abs(a0[nm]).max(), nm
    assert np.abs(g1 - g0).max() > 0, "both runs took the same path: the switch did not reach the engine"
    assert np.abs(g1 - g0).max() <= 5e-3 * np.abs(g0).max()


def test_norm_backward_statistics_in_the_small_volume_dgrad_epilogue_match_the_separate_pass(tmp_path):
    """The same epilogue in k_mfma_conv_small (the dgrads of the 16^3 level -- 8^3 and below run on the split-K kernels of
    kernels_mfma_deep.hip, which do the whole norm backward; UNET_NO_DGRAD_BNSTATS_SMALL=1 keeps the separate k_norm_bwd_stats8
    launches): the default architecture at 64^3 has its 64-channel level at 16^3.  The first such layer
    in backward order is decode2.3's dgrad (statistics of decode2.1's norm): that norm's parameter gradients agree to summation-order
    noise, everything computed before it (and the heads, which read no dL/d(raw)) exactly; the rest is bounded loosely (bf16 roundings
    of du flip downstream)."""
    g1, names = _grads_in_fresh_process(tmp_path, "small_fused", 64, {}, arch="default")
    g0, _ = _grads_in_fresh_process(tmp_path, "small_separate", 64, {"UNET_NO_DGRAD_BNSTATS_SMALL": "1"}, arch="default")
    a1, a0 = _by_name(g1, names), _by_name(g0, names)
    assert np.abs(g1 - g0).max() > 0, "both runs took the same path: the switch did not reach the engine"
    for nm in a0:
        if nm.startswith(("decode0.", "decode1.", "decode2.3.", "decode2.4.", "output")):
            assert np.array_equal(a1[nm], a0[nm]), nm            # upstream of the first fused layer
    for nm in ("decode2.1.weight", "decode2.1.bias"):
        assert np.abs(a1[nm] - a0[nm]).max() <= 1e-5 * np.abs(a0[nm]).max(), nm
    assert np.abs(g1 - g0).max() <= 2e-2 * np.abs(g0).max()


def test_stride2_dgrad_statistics_epilogue_and_kernels_match_the_separate_pass_and_the_halo_tile_kernels(tmp_path):
    """kernels_mfma_s2.hip in the network: the stride-2 conv reads the skip tensor, whose gradient has two writers -- its dgrad
    (k_s2_scatter) ACCUMULATES into it (old values by LDS-DMA) and, as the last writer, leaves the norm backward's statistics
    (UNET_NO_DGRAD_BNSTATS=1: separate k_norm_bwd_stats8 pass; UNET_NO_SLIDING_WINDOW=1: the halo-tile kernels for every conv /
    conv_trans op).  At 64^3 the small U-Net's coarse level is 32^3: all four new kernels run.  Each variant in a fresh process (the
    switches are read once).  The statistics agree to summation order, so encode0.4's norm parameters do; the rest sees bf16 rounding flips."""
    g1, names = _grads_in_fresh_process(tmp_path, "s2_fused", 64, {})
    g0, _ = _grads_in_fresh_process(tmp_path, "s2_separate", 64, {"UNET_NO_DGRAD_BNSTATS": "1"})
    g2, _ = _grads_in_fresh_process(tmp_path, "s2_old", 64, {"UNET_NO_SLIDING_WINDOW": "1"})
    a1, a0, a2 = _by_name(g1, names), _by_name(g0, names), _by_name(g2, names)
    assert np.abs(g1 - g0).max() > 0 and np.abs(g1 - g2).max() > 0, "the switches did not reach the engine"
    for nm in ("encode0.4.weight", "encode0.4.bias"):      # fp32 sums over 64^3 voxels with cancellation, grouped differently: measured 7e-5
        assert np.abs(a1[nm] - a0[nm]).max() <= 5e-4 * np.abs(a0[nm]).max(), nm
    assert np.abs(g1 - g0).max() <= 5e-3 * np.abs(g0).max()
    # against the halo-tile kernels: same arithmetic in another summation order (fp32 accumulators, then bf16 roundings downstream)
    # (floor of 1e-2 of the largest gradient: a conv bias in front of a norm has an analytically zero gradient -- both sides hold ~3e-5 of
    # rounding noise there, of which only the size is comparable)
    for nm in a2:
        assert np.abs(a1[nm] - a2[nm]).max() <= 2e-2 * max(np.abs(a2[nm]).max(), 1e-2 * np.abs(g2).max()), nm


def test_deep_level_split_k_kernels_match_the_halo_tile_kernels_in_the_network(tmp_path):
    """kernels_mfma_deep.hip in the network: at the deep levels the contractions run split over K, finished by the block that arrives
    last, with the norm layer behind the conv (forward: statistics, running statistics, activated copy) or in front of it (backward:
    statistics, affine gradients, dL/d(raw)) in that epilogue.  UNET_NO_DEEP_KERNELS=1 keeps the halo-tile kernels and the separate norm
    launches there.  A two-level network of 32 / 64 channels: at 8^3 its coarse level is 4^3 (the 27-tap kinds with both norm epilogues,
    one dgrad accumulating into the skip gradient), at 16^3 it is 8^3 (the short kinds only: conv_trans forward / dgrad, stride-2 dgrad).
    The two paths do the same arithmetic in another summation order and bf16 roundings downstream amplify that (these are tiny
    volumes: a few % of a tensor's size), so each is measured against the fp32 ENGINE (other kernels, fp32 storage) on the same
    sample: the new path's distance to it must be the old path's, tensor by tensor.  And the default architecture at 32^3 (levels 3..5:
    4^3 x 128 channels ... 1^3 x 256, every shape of the path) is reproducible run to run: fixed summation order whichever block
    arrives last."""
    for n in (8, 16):
        g1, names = _grads_in_fresh_process(tmp_path, "deep_on_%d" % n, n, {}, arch="deep2")
        g0, _ = _grads_in_fresh_process(tmp_path, "deep_off_%d" % n, n, {"UNET_NO_DEEP_KERNELS": "1"}, arch="deep2")
        gf, _ = _grads_in_fresh_process(tmp_path, "fp32_%d" % n, n, {}, arch="deep2", dtype="fp32")
        assert np.abs(g1 - g0).max() > 0, "the switch did not reach the engine"
        a1, a0, af = _by_name(g1, names), _by_name(g0, names), _by_name(gf, names)
        floor = 1e-2 * np.abs(gf).max()
        for nm in af:
            e1, e0, ref = np.abs(a1[nm] - af[nm]).max(), np.abs(a0[nm] - af[nm]).max(), max(np.abs(af[nm]).max(), floor)
            assert e1 <= 2.0 * e0 + 5e-3 * ref, (n, nm, e1 / ref, e0 / ref)
            assert e1 <= 0.6 * ref, (n, nm, e1 / ref)      # (a wrong tap or channel is off by O(1) on most elements; bf16 with 64-voxel norms: the old path measures 0.17-0.35 on single elements of encode1.3.weight at 8^3)
    d1, _ = _grads_in_fresh_process(tmp_path, "deep_default", 32, {}, arch="default")
    d1b, _ = _grads_in_fresh_process(tmp_path, "deep_default_again", 32, {}, arch="default")
    d0, _ = _grads_in_fresh_process(tmp_path, "deep_default_off", 32, {"UNET_NO_DEEP_KERNELS": "1"}, arch="default")
    assert np.array_equal(d1, d1b) and np.abs(d1 - d0).max() > 0
    # (the six-level network amplifies the summation-order noise of its 1^3 .. 4^3 levels: only the sizes are comparable there)
    assert abs(np.linalg.norm(d1) - np.linalg.norm(d0)) <= 5e-2 * np.linalg.norm(d0)


_DEEP_BNORM_CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import unet_studio_amd as U
arch = ("conv32,ks3,stride1+bnorm,leaky_relu+conv32,ks3,stride1+bnorm,leaky_relu\n"
        "conv64,ks3,stride2+bnorm,leaky_relu+conv64,ks3,stride1+bnorm,leaky_relu+conv_trans32,ks2,stride2\n"
        "conv32,ks3,stride1+bnorm,leaky_relu+conv32,ks3,stride1+bnorm,leaky_relu+conv6,ks1,stride1")
m = U.UNet3d(1, 6, arch, device="cuda:0", dtype="bf16", seed=0)
feed = U.SyntheticVolumes(1, 6, (8, 8, 8), "cuda:0", cache=2)
for k in range(2):                       # two training micro-steps: the running statistics move twice
    x, t = feed(k)
    m.forward_backward(x, t)
torch.cuda.synchronize()
bufs = np.concatenate([b.detach().float().cpu().numpy().ravel() for b in m.buffers()])
m.eval()
with torch.no_grad():
    outs = m.forward(feed(0)[0])         # eval(): bnorm normalises with the running statistics (train.cpp:834-852)
torch.cuda.synchronize()
np.savez(sys.argv[2], grads=m.flat_grads.cpu().numpy(), bufs=bufs, **{"out%d" % k: o.float().cpu().numpy() for k, o in enumerate(outs)})
"""


def test_deep_level_kernels_with_bnorm_running_statistics_and_eval_forward(tmp_path):
    """The norm epilogue of k_deep_conv also owns BatchNorm3d's bookkeeping (unet.cpp:80-84): in training it updates the running mean /
    unbiased variance (momentum 0.1), in eval() it normalises with them (scale / shift from the running statistics, no batch
    statistics).  A two-level bnorm network at 8^3 (coarse level 4^3: the fused path), two training micro-steps, then an eval forward --
    against the same run with UNET_NO_DEEP_KERNELS=1 (k_norm_finalize_apply8 / k_norm_eval): buffers to summation order, eval logits and
    gradients to bf16 rounding flips."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env_extra in (("deep", {}), ("halo", {"UNET_NO_DEEP_KERNELS": "1"})):
        path = str(tmp_path / ("bn_%s.npz" % tag))
        out = subprocess.run([sys.executable, "-c", _DEEP_BNORM_CHILD, root, path], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env_extra))
        assert out.returncode == 0, out.stderr[-2000:]
        res[tag] = np.load(path)
    a, b = res["deep"], res["halo"]
    assert np.abs(a["grads"] - b["grads"]).max() > 0, "the switch did not reach the engine"
    # running statistics: fp32 sums over 64 voxels in another order, then the values downstream of the first fused layer move by bf16
    # rounding flips of its output
    assert np.abs(a["bufs"] - b["bufs"]).max() <= 2e-2 * np.abs(b["bufs"]).max()
    assert np.isfinite(a["bufs"]).all() and np.isfinite(a["grads"]).all()
    outs = [k for k in a.files if k.startswith("out")]
    assert "out0" in outs
    for k in outs:
        assert rel(a[k], b[k]) < 3e-2, k
    assert abs(np.linalg.norm(a["grads"]) - np.linalg.norm(b["grads"])) <= 5e-2 * np.linalg.norm(b["grads"])


def test_halo_tile_kernels_still_match_the_oracle():
    """UNET_NO_SLIDING_WINDOW=1 (mfma_util.h) is the documented fallback of every kernel that counts its vector-memory operations by
    hand (k_mfma_conv_z / _z16 / _z32, k_mfma_wgrad_z / _zd, k_s2_*): all shapes then run on the halo-tile kernels k_mfma_conv_p /
    k_mfma_conv_small / k_mfma_wgrad.  Those must stay correct for it to BE a fallback: the bf16 op cases run again in a child with
    the switch set (read once per process)."""
    import subprocess
    import sys
    env = dict(os.environ, UNET_NO_SLIDING_WINDOW="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                          "-p", "no:cacheprovider", "-k", "(test_conv3d_ops or test_convt_ops) and bf16"], capture_output=True, text=True, timeout=900,
                         env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_one_stream_equals_two_streams_bit_for_bit(tmp_path):
    """UNET_NO_SIDE_STREAM=1 (engine.cpp; the diagnostic that profiles/stretch.py and the per-op profiler build on) puts the filter
    packs, the coarse levels' loss and every weight gradient on the caller's stream.  The two streams never write the same buffer and
    every reduction has a fixed order, so the gradients must be IDENTICAL -- fresh processes, the switch is read once."""
    g1, _ = _grads_in_fresh_process(tmp_path, "two_streams", 32, {}, arch="default")
    g0, _ = _grads_in_fresh_process(tmp_path, "one_stream", 32, {"UNET_NO_SIDE_STREAM": "1"}, arch="default")
    assert np.array_equal(g1, g0)


def test_weight_gradients_in_the_polite_launch_configuration():
    """The engine launches the weight gradients of its side stream "politely" (engine.cpp: choose_polite -- 4-wave blocks, one per CU,
    the sliding-window kernel in (2 x 1) / (1 x 2) tile pairs of 4 rows instead of the 8-wave pair blocks): UNET_OP_POLITE=1 gives the
    op-level entry point the same configuration, and every conv case of this file must still match the oracle.  The switch is read
    once per process: the cases run again in a child."""
    import subprocess
    import sys
    env = dict(os.environ, UNET_OP_POLITE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                          "-p", "no:cacheprovider", "-k", "test_conv3d_ops and bf16"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_first_conv_weight_gradient_with_the_norm_backward_pass_fused_equals_the_separate_pass(tmp_path):
    """dL/d(raw output) of the network's first conv is read by nothing but its weight gradient, so that kernel applies the norm
    backward's element-wise pass while it stages its tiles (NormBwdFuse, engine.cpp) instead of reading a tensor a separate pass wrote;
    UNET_NO_FIRST_WGRAD_FUSE=1 keeps the separate pass.  The fused form rounds to bf16 exactly where the pass stores, so at 32^3 (both
    routes take k_norm_bwd_finalize) every gradient is bit-identical; at 16^3 the separate route sums the statistics rows in the
    fused finalize's order, so the first conv's gradients agree to summation-order noise and everything else exactly."""
    g1, names = _grads_in_fresh_process(tmp_path, "fused32", 32, {})
    g0, _ = _grads_in_fresh_process(tmp_path, "separate32", 32, {"UNET_NO_FIRST_WGRAD_FUSE": "1"})
    assert np.array_equal(g1, g0)
    assert np.abs(_by_name(g1, names)["encode0.0.weight"]).max() > 0
    h1, names = _grads_in_fresh_process(tmp_path, "fused16", 16, {})
    h0, _ = _grads_in_fresh_process(tmp_path, "separate16", 16, {"UNET_NO_FIRST_WGRAD_FUSE": "1"})
    a1, a0 = _by_name(h1, names), _by_name(h0, names)
    for nm in a0:
        if nm.startswith("encode0.0.") or nm.startswith("encode0.1."):
            assert np.abs(a1[nm] - a0[nm]).max() <= 1e-4 * max(np.abs(a0[nm]).max(), 1e-30), nm
        else:
            assert np.array_equal(a1[nm], a0[nm]), nm


ARCH_RESUME = ("conv16,ks3,stride1+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu\n"
               "conv32,ks3,stride2+norm,leaky_relu+conv32,ks3,stride1+norm,leaky_relu+conv_trans16,ks2,stride2\n"
               "conv16,ks3,stride1+norm,leaky_relu+conv16,ks3,stride1+norm,leaky_relu+conv4,ks1,stride1")


def _trainer(dt, batch=2, model=None):
    m = model if model is not None else U.UNet3d(1, 4, ARCH_RESUME, device=DEV, dtype=dt, seed=0)
    src = U.SyntheticVolumes(1, 4, (16, 16, 16), DEV, cache=8)
    return m, U.Trainer(m, U.TrainingParam(batch_size=batch, epoch=100, learning_rate=0.05), lambda i: src(i % 8))


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_resume_from_network_file_and_optimizer_file_equals_uninterrupted(tmp_path, dt):
    """train.cpp:787 saves <model>.nz + <model>.nz.opt every 100 epochs, :945-957 loads the optimizer on restart: two optimizer steps,
    checkpoint, a FRESH model that loads both and continues for two steps == four uninterrupted steps, bit for bit (the fused
    update's momentum is what the .opt file has to carry); without the .opt file the result differs (so the test is not vacuous)."""
    from unet_studio_amd import nz
    m4, t4 = _trainer(dt)
    for _ in range(4):
        t4.step()
    m2, t2 = _trainer(dt)
    for _ in range(2):
        t2.step()
    path = str(tmp_path / "net.nz")
    assert nz.save_to_file(m2, path) and m2.save_optimizer(path + ".opt")
    torch.cuda.synchronize()

    def resumed(with_opt):
        m = nz.load_from_file(path, lambda i, o, a: U.UNet3d(i, o, a, device=DEV, dtype=dt))
        m.create_optimizer(0.05)
        if with_opt:
            assert m.load_optimizer(path + ".opt"), m.error_msg
        _, t = _trainer(dt, model=m)
        t.cur_epoch = 2          # the epoch counter lives in the caller (train.cpp:562 cur_epoch), the poly schedule reads it
        for _ in range(2):
            t.step()
        torch.cuda.synchronize()
        return m.flat_params.clone()

    assert torch.equal(resumed(True), m4.flat_params), "resumed run differs from the uninterrupted one"
    assert not torch.equal(resumed(False), m4.flat_params), "momentum made no difference: the resume test is vacuous"
    # a file of another architecture is refused with the reference's message prefix, not loaded by size coincidence
    other = U.UNet3d(1, 4, ARCH_RESUME.replace("conv32", "conv48"), device=DEV, dtype=dt, seed=0)
    other.create_optimizer(0.05)
    assert not other.load_optimizer(path + ".opt") and other.error_msg.startswith("cannot load optimizer")


@pytest.mark.parametrize("dt", ["bf16"])
def test_micro_steps_reusing_the_filter_packs_equal_repacking(dt):
    """UNET_MODE_PACKS_CURRENT (include/unet_hip.h): micro-steps 2.. of one optimizer step skip the filter repack because the
    parameters only change at the end of the step (train.cpp:604-606,765).  Three steps of batch 3 with and without the reuse must
    give bit-identical parameters and loss statistics."""
    ma, ta = _trainer(dt, batch=3)
    mb, tb = _trainer(dt, batch=3)
    tb.packs_reuse = False
    assert ta.packs_reuse
    for _ in range(3):
        sa, sb = ta.step().clone(), tb.step().clone()
        assert torch.equal(sa, sb)
    torch.cuda.synchronize()
    assert torch.equal(ma.flat_params, mb.flat_params)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_filter_packs_made_behind_the_update_equal_the_forwards_own(dt):
    """unet_pack_filters (include/unet_hip.h): a trainer may make the workspace's filter packs right behind the update and give the next
    step's first forward UNET_MODE_PACKS_CURRENT.  Three steps with Trainer.pack_after_update must give bit-identical parameters and
    loss statistics; the fp32 engine has no batched pack: pack_filters reports so and the forward repacks as before."""
    ma, ta = _trainer(dt, batch=2)
    mb, tb = _trainer(dt, batch=2)
    ta.pack_after_update = True
    assert not tb.pack_after_update
    for _ in range(3):
        sa, sb = ta.step().clone(), tb.step().clone()
        assert torch.equal(sa, sb)
        assert (ta._packed_size == (16, 16, 16)) == (dt == "bf16")
    torch.cuda.synchronize()
    assert torch.equal(ma.flat_params, mb.flat_params)


@pytest.mark.parametrize("lane_cus", [0, 12])
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_micro_steps_in_flight_equal_the_sequential_order(dt, lane_cus, monkeypatch):
    """Two micro-steps of one optimizer step side by side on one GPU (Trainer.in_flight = 2: two lanes = two streams, two plans, two
    workspaces; every micro-step writes a gradient buffer of its own; unet_sum_buffers adds them in micro-step order) must give the
    parameters of the sequential order BIT FOR BIT: ((g0 + g1) + g2) + ... is what one accumulating buffer holds (train.cpp:604-606,
    756-761).  Batch 5 = an odd count (lanes of 3 and 2 micro-steps), three steps (the buffers are cleared and reused)."""
    if lane_cus:   # the lanes (stream and plan side stream) on disjoint CU ranges of every XCD (unet_stream_create_cu_range); such streams
        monkeypatch.setenv("UNET_LANE_CUS", str(lane_cus))   # synchronize with the NULL stream, so the trainer runs on a stream of torch's pool
    ma, ta = _trainer(dt, batch=5)
    mb, tb = _trainer(dt, batch=5)
    ta.in_flight, tb.in_flight = 2, 1
    torch.cuda.synchronize()
    with torch.cuda.stream(torch.cuda.Stream(DEV)):
        _in_flight_steps(ma, ta, mb, tb)
    torch.cuda.synchronize()


def _in_flight_steps(ma, ta, mb, tb):
    for _ in range(3):
        sa, sb = ta.step().clone(), tb.step().clone()
        assert torch.allclose(sa, sb, rtol=1e-6, atol=1e-7)
    torch.cuda.synchronize()
    assert ta._lanes is not None and tb._lanes is None
    assert torch.equal(ma.flat_params, mb.flat_params)
    assert float(ma.flat_grads.abs().max()) == 0.0 and all(float(g.abs().max()) == 0.0 for g in ta._gbufs)   # zero_grad reached every buffer


def test_backward_refuses_a_gradient_buffer_for_the_input():
    """dL/dx is not computed (the reference's input is a leaf without requires_grad, train.cpp:619-628): a non-NULL grad_x is an error with
    a message, not a write through a buffer the plan never laid out"""
    m = U.UNet3d(1, 4, ARCH_RESUME, device=DEV, dtype="bf16", seed=0)
    x, t = U.SyntheticVolumes(1, 4, (16, 16, 16), DEV, cache=1)(0)
    plan = m.plan_for(x.shape[2:]); ws = m._workspace(plan)
    outs, losses, gouts = m._run_forward_loss(plan, ws, x, t, True, True, True, 0)
    with pytest.raises(U.UNetError, match="grad_x"):
        m._run_backward(plan, ws, gouts, grad_x=torch.empty_like(x))
    m._run_backward(plan, ws, gouts)
    torch.cuda.synchronize()
    assert float(m.flat_grads.abs().max()) > 0


def test_sum_buffers_is_the_sequential_accumulation():
    """unet_sum_buffers (include/unet_hip.h): ((b0 + b1) + b2) + ... in fp32, inputs cleared on request, odd tail, argument checks"""
    n = 4 * 1000 + 3
    g = torch.Generator(device="cpu").manual_seed(5)
    bufs = [(torch.randn(n + 1, generator=g) * 10 ** k).to(DEV)[:n] for k in range(5)]      # magnitudes 1 .. 1e4: the order matters
    bufs = [torch.empty(n + 4, device=DEV)[:n].copy_(b) for b in bufs]                         # 16-byte aligned views
    want = bufs[0].clone()
    for b in bufs[1:]:
        want = want + b
    out = torch.empty(n + 4, device=DEV)[:n]
    E.check(E.lib.unet_sum_buffers(E.ptr_array([b.data_ptr() for b in bufs]), 5, out.data_ptr(), n, 1, stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    assert all(float(b.abs().max()) == 0.0 for b in bufs)
    assert E.lib.unet_sum_buffers(E.ptr_array([out.data_ptr() + 4]), 1, out.data_ptr(), 8, 0, stream()) != 0     # misaligned input
    assert E.lib.unet_sum_buffers(E.ptr_array([out.data_ptr()] * 65), 65, out.data_ptr(), 8, 0, stream()) != 0   # too many buffers
