"""CPU: `.nz` network files (unet-studio_amd/nz.py and csrc/nz_io.cpp; reference main.cpp:157-233 over TIPL's gz_mat container).
PARITY UNPINNED against the reference (no .nz file / TIPL source in the tree): pinned here are the MATLAB Level-4 record format
(known answer assembled by hand from its public description), the round trip, the reference's error messages, and that the
Python and the C++ host read each other's files."""
import gzip
import os
import struct
import subprocess
import threading

import numpy as np
import pytest
import torch

import unet_studio_amd as U
from unet_studio_amd import nz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "unet-studio_amd")
ARCH = ("conv8,ks3,stride1+norm,leaky_relu\nconv16,ks3,stride2+norm,leaky_relu+conv_trans8,ks2,stride2\n"
        "conv8,ks3,stride1+norm,leaky_relu+conv3,ks1,stride1")


class HostModel:
    """the fields and calls load_from_file / save_to_file touch (unet.hpp:16-40), with CPU tensors in parameters() order"""

    def __init__(self, in_count, out_count, architecture):
        self.in_count, self.out_count, self.architecture = in_count, out_count, architecture
        plan = U.Plan(architecture, in_count, out_count, (32, 32, 32))     # parameter order / shapes from the engine's DSL parser
        g = torch.Generator().manual_seed(1)
        self._p = [torch.randn(s, generator=g) for s in plan.param_shapes]
        self.fov_strategy, self.preproc, self.postproc, self.orientation, self.error_msg = "align_top", "", "softmax+create_mask+argmax", "", ""
        self.voxel_size, self.dim = (1.0, 1.0, 1.0), (192, 224, 192)
        self.testing_errors, self.training_errors, self.single_component_label = [], [], []
        self.error_mutex, self.training = threading.Lock(), False

    def parameters(self):
        return self._p

    def train(self):
        self.training = True

    def load_parameters(self, arrays):
        assert len(arrays) == len(self._p)
        self._p = [torch.from_numpy(np.array(a, np.float32)).reshape(p.shape) for a, p in zip(arrays, self._p)]


def test_level4_record_known_answer():
    """MAT-File Level 4: x = [1 2 3; 4 5 6] as doubles is {0, 2, 3, 0, 2} 'x\\0' then 1 4 2 5 3 6 (column-major)"""
    raw = struct.pack("<5i", 0, 2, 3, 0, 2) + b"x\0" + struct.pack("<6d", 1, 4, 2, 5, 3, 6)
    raw += struct.pack("<5i", 51, 1, 5, 0, 5) + b"name\0" + b"hello"           # text in uint8: P = 5, T = 1
    raw += struct.pack("<5i", 10, 3, 1, 0, 2) + b"v\0" + struct.pack("<3f", 0.5, -1.0, 2.0)
    recs = nz.read_records(raw)
    assert list(recs) == ["x", "name", "v"]
    assert np.array_equal(recs["x"][0], [[1, 2, 3], [4, 5, 6]]) and recs["x"][0].dtype == np.float64
    assert nz._text(recs["name"]) == "hello" and recs["name"][1]
    assert np.array_equal(recs["v"][0].reshape(-1), np.array([0.5, -1.0, 2.0], np.float32))
    import io
    buf = io.BytesIO()
    nz.write_record(buf, "x", np.array([[1, 2, 3], [4, 5, 6]], np.float64))
    assert buf.getvalue() == raw[:20 + 2 + 48]
    with pytest.raises(nz.NzError):
        nz.read_records(struct.pack("<5i", 1000, 1, 1, 0, 2) + b"x\0" + b"\0" * 8)    # big-endian marker: refused


def _model():
    m = HostModel(2, 3, ARCH)
    m.voxel_size, m.dim = (0.5, 0.75, 1.25), (96, 112, 80)
    m.preproc, m.orientation = "normalize", "LPS"
    m.training_errors = [0.5, 0.25, 0.125, 0.4, 0.2, 0.1]
    m.testing_errors = [0.6, 0.3, 0.15, 0.5, 0.25, 0.12]
    return m


def test_round_trip_and_record_order(tmp_path):
    m = _model()
    f = str(tmp_path / "net.nz")
    assert nz.save_to_file(m, f)
    recs = nz.read_records(gzip.open(f, "rb").read())
    head = ["channels", "architecture", "dimension", "voxel_size", "fov_strategy", "preproc", "orientation", "postproc", "training_errors", "testing_errors"]
    assert list(recs)[:10] == head and list(recs)[10:] == ["tensor%d" % i for i in range(len(m.parameters()))]    # main.cpp:212-231
    assert recs["training_errors"][0].shape == (3, 2)                                                             # write(name, v, 3)
    w = m.parameters()[0]
    assert recs["tensor0"][0].shape == (w.numel() // w.shape[0], w.shape[0])                                      # rows = numel / size(0)
    r = nz.load_from_file(f, HostModel)
    assert (r.in_count, r.out_count, r.architecture) == (2, 3, ARCH)
    assert r.dim == m.dim and np.allclose(r.voxel_size, m.voxel_size) and r.training
    assert (r.fov_strategy, r.preproc, r.orientation, r.postproc) == (m.fov_strategy, m.preproc, m.orientation, m.postproc)
    assert np.allclose(r.training_errors, m.training_errors) and np.allclose(r.testing_errors, m.testing_errors)
    for a, b in zip(r.parameters(), m.parameters()):
        assert torch.equal(a, b)


def test_reference_error_messages(tmp_path):
    m = _model()
    f = str(tmp_path / "net.nz")
    nz.save_to_file(m, f)
    raw = gzip.open(f, "rb").read()
    recs = nz.read_records(raw)
    # drop the last tensor -> "tensor size mismatch at tensor<i> 0 not the expected of size n" (main.cpp:198-200)
    last = "tensor%d" % (len(m.parameters()) - 1)
    with gzip.open(f, "wb") as g:
        for k, (a, t) in recs.items():
            if k != last:
                nz.write_record(g, k, a, text=t)
    with pytest.raises(nz.NzError, match="tensor size mismatch at " + last):
        nz.load_from_file(f, HostModel)
    with gzip.open(f, "wb") as g:
        nz.write_record(g, "architecture", recs["architecture"][0], text=True)
    with pytest.raises(nz.NzError, match="invalid format"):
        nz.load_from_file(f, HostModel)
    # a tensor in TIPL's sloped encoding is refused, not guessed
    with gzip.open(f, "wb") as g:
        for k, (a, t) in recs.items():
            if k == "tensor0":
                nz.write_record(g, k, np.zeros(a.shape, np.int16))
                nz.write_record(g, k + ".slope", np.ones((1, 1), np.float32))
            else:
                nz.write_record(g, k, a, text=t)
    with pytest.raises(nz.NzError, match="sloped"):
        nz.load_from_file(f, HostModel)
    with pytest.raises(nz.NzError):
        nz.load_from_file(str(tmp_path / "missing.nz"), HostModel)


def test_cpp_host_reads_and_writes_the_same_files(tmp_path):
    exe = os.path.join(PKG, "test_nz_io")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(PKG, "csrc", "build_host.sh")])
    m = _model()
    fin, fout = str(tmp_path / "py.nz"), str(tmp_path / "cpp.nz")
    assert nz.save_to_file(m, fin)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = PKG + ":" + os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, fin, fout], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    assert "in 2 out 3 params %d" % len(m.parameters()) in r.stdout and "errors 6 6" in r.stdout
    assert "dim: 96 112 80 reso: 0.5 0.75 1.25" in r.stdout and "preproc: normalize" in r.stdout      # get_info, unet.cpp:279-291
    a, b = nz.read_records(gzip.open(fin, "rb").read()), nz.read_records(gzip.open(fout, "rb").read())
    assert list(a) == list(b)
    for k in a:
        assert a[k][1] == b[k][1] and a[k][0].shape == b[k][0].shape and np.array_equal(a[k][0], b[k][0]), k
    back = nz.load_from_file(fout, HostModel)
    for x, y in zip(back.parameters(), m.parameters()):
        assert torch.equal(x, y)


def _corrupt_files(tmp_path):
    """crafted / damaged network files: a header that declares 2^31-1 x 2^31-1 doubles, a payload cut short, a negative type code,
    channels with one element, garbage that is not gzip"""
    import struct
    m = _model()
    good = str(tmp_path / "good.nz")
    assert nz.save_to_file(m, good)
    raw = gzip.open(good, "rb").read()
    files = {}

    def put(name, payload):
        f = str(tmp_path / (name + ".nz"))
        with gzip.open(f, "wb") as g:
            g.write(payload)
        files[name] = f
    put("huge", struct.pack("<5i", 0, 2 ** 31 - 1, 2 ** 31 - 1, 0, 9) + b"channels\0" + b"\0" * 64)
    put("truncated", raw[: len(raw) // 2])
    put("negative_type", struct.pack("<5i", -10, 1, 2, 0, 9) + b"channels\0" + b"\0" * 8)
    put("short_channels", struct.pack("<5i", 20, 1, 1, 0, 9) + b"channels\0" + struct.pack("<i", 1) + raw[raw.index(b"architecture") - 20:])
    f = str(tmp_path / "notgzip.nz")
    open(f, "wb").write(b"this is not a gzip stream")
    files["notgzip"] = f
    return files


def test_damaged_files_are_refused_with_a_message_by_the_python_reader(tmp_path):
    for name, f in _corrupt_files(tmp_path).items():
        with pytest.raises(nz.NzError):
            nz.load_from_file(f, HostModel)


def test_damaged_files_are_refused_with_a_message_by_the_cpp_reader(tmp_path):
    """load_from_file's contract (main.cpp:157-206) is `false + error_msg`, never an exception: a crafted header must not reach
    std::vector::resize with 2^65 bytes, a short stream must fail on the read, the gzFile must not leak (ASan run: sanitize_host.sh)"""
    exe = os.path.join(PKG, "test_nz_io")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(PKG, "csrc", "build_host.sh")])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = PKG + ":" + os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")
    for name, f in _corrupt_files(tmp_path).items():
        r = subprocess.run([exe, f, str(tmp_path / "out.nz")], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and "LOAD FAILED" in r.stdout, (name, r.returncode, r.stdout[-300:], r.stderr[-300:])
