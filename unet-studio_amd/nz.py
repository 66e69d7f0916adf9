"""`.nz` network files: load_from_file / save_to_file of the reference (main.cpp:157-233).

The reference delegates the container to TIPL (`tipl::io::gz_mat_read` / `gz_mat_write`), which is neither vendored with it nor
present here, so this is a restatement from the public description of the container -- a gzip stream of MATLAB Level-4 MAT
records -- with the record set, order and shapes of main.cpp:207-233.  PARITY UNPINNED: no `.nz` file and no TIPL source exist in
the tree to check the byte layout against.  What is pinned is the Level-4 record format itself
(known-answer test in tests/test_nz.py) and the round trip.

Level-4 record: five little-endian int32 {type, mrows, ncols, imagf, namlen}, the NUL-terminated name, then mrows*ncols elements
column-major.  type = M*1000 + O*100 + P*10 + T with M = 0 (little endian), P = 0 double / 1 float / 2 int32 / 3 int16 /
4 uint16 / 5 uint8, T = 0 numeric / 1 text.

Records (main.cpp:212-231): channels (int32 x2: in, out), architecture (text), dimension (3 x uint32 stored as int32), voxel_size
(3 x float), fov_strategy / preproc / orientation / postproc (text), training_errors / testing_errors (float, 3 rows), then
tensor0..tensorN-1 in parameters() order (= unet_plan_param_* order), each float with rows = numel / size(0), cols = size(0).

NOT supported: the reference writes the tensors through `tipl::io::sloped` (main.cpp:223-229: a TIPL-defined quantised / slope-
intercept encoding, `apply_slope`, `min_size_for_mask_slope`).  Its layout is TIPL's and unknown here; this module always writes
plain float tensors and refuses to read a tensor record that is not plain float / double (clear error naming TIPL), rather than
guessing a dequantisation.
"""
import gzip
import struct

import numpy as np

_P = {0: np.dtype("<f8"), 1: np.dtype("<f4"), 2: np.dtype("<i4"), 3: np.dtype("<i2"), 4: np.dtype("<u2"), 5: np.dtype("u1")}
_PCODE = {np.dtype("<f8"): 0, np.dtype("<f4"): 1, np.dtype("<i4"): 2, np.dtype("<i2"): 3, np.dtype("<u2"): 4, np.dtype("u1"): 5}


class NzError(RuntimeError):
    pass


def write_record(f, name, array, text=False):
    """one Level-4 record; `array` is written column-major as (rows, cols)"""
    a = np.asarray(array)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.ndim != 2:
        raise NzError("record %s: Level-4 matrices are two-dimensional" % name)
    dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
    if np.dtype(dt) not in _PCODE:
        raise NzError("record %s: unsupported element type %s" % (name, a.dtype))
    nm = name.encode() + b"\0"
    f.write(struct.pack("<5i", _PCODE[np.dtype(dt)] * 10 + (1 if text else 0), a.shape[0], a.shape[1], 0, len(nm)))
    f.write(nm)
    f.write(np.asfortranarray(a.astype(dt, copy=False)).tobytes(order="F"))


def read_records(data):
    """bytes of the (already decompressed) stream -> {name: (array (rows, cols), is_text)} in file order"""
    out, off, n = {}, 0, len(data)
    while off < n:
        if off + 20 > n:
            raise NzError("truncated record header at byte %d" % off)
        typ, rows, cols, imagf, namlen = struct.unpack_from("<5i", data, off)
        off += 20
        m, rest = divmod(typ, 1000)
        o, rest = divmod(rest, 100)
        p, t = divmod(rest, 10)
        if m != 0 or o != 0 or p not in _P or t not in (0, 1) or rows < 0 or cols < 0 or namlen < 1 or imagf not in (0, 1):
            raise NzError("not a little-endian Level-4 MAT record at byte %d (type %d)" % (off - 20, typ))
        name = data[off:off + namlen].split(b"\0")[0].decode("latin-1")
        off += namlen
        nbytes = rows * cols * _P[p].itemsize * (2 if imagf else 1)
        if off + nbytes > n:
            raise NzError("record %s is truncated" % name)
        a = np.frombuffer(data, dtype=_P[p], count=rows * cols, offset=off).reshape((rows, cols), order="F")
        off += nbytes
        out[name] = (a, t == 1)
    return out


def _text(rec):
    a, _ = rec
    return bytes(np.asarray(a, dtype=np.uint8).reshape(-1, order="F")).split(b"\0")[0].decode("latin-1")


def save_to_file(model, file_name):
    """bool save_to_file(UNet3d& model, const char* file_name) -- main.cpp:207-233 (tensors as plain float, see the module text)"""
    try:
        with gzip.open(file_name, "wb") as f:
            write_record(f, "channels", np.array([model.in_count, model.out_count], np.int32))
            write_record(f, "architecture", np.frombuffer(model.architecture.encode("latin-1"), np.uint8), text=True)
            write_record(f, "dimension", np.array(list(model.dim), np.int32))
            write_record(f, "voxel_size", np.array(list(model.voxel_size), np.float32))
            for key in ("fov_strategy", "preproc", "orientation", "postproc"):
                write_record(f, key, np.frombuffer(getattr(model, key).encode("latin-1"), np.uint8), text=True)
            for key in ("training_errors", "testing_errors"):
                e = np.asarray(getattr(model, key), np.float32)
                e = e[: (e.size // 3) * 3]
                write_record(f, key, e.reshape((3, -1), order="F"))
            for i, p in enumerate(model.parameters()):
                a = p.detach().to("cpu").contiguous().numpy().astype(np.float32, copy=False)
                cols = a.shape[0]
                write_record(f, "tensor%d" % i, a.reshape(-1).reshape((a.size // cols, cols), order="F"))
        return True
    except OSError as e:
        model.error_msg = str(e)
        return False


def load_from_file(file_name, make_model):
    """bool load_from_file(UNet3d& model, const char* file_name) -- main.cpp:157-206.  make_model(in_count, out_count, architecture)
    builds the UNet3d (the reference assigns `model = UNet3d(param[0], param[1], architecture)`); returns the model.
    Raises NzError with the reference's messages ("invalid format", "tensor size mismatch at tensor<i> ...")."""
    try:
        with gzip.open(file_name, "rb") as f:
            recs = read_records(f.read())
    except (OSError, EOFError) as e:
        raise NzError(str(e))
    if "channels" not in recs or "architecture" not in recs:
        raise NzError("invalid format")
    ch = np.asarray(recs["channels"][0]).reshape(-1)
    if ch.size < 2:
        raise NzError("invalid format")
    arch = _text(recs["architecture"])
    model = make_model(int(ch[0]), int(ch[1]), arch)
    if "dimension" not in recs or "voxel_size" not in recs:
        raise NzError("invalid format")
    if np.asarray(recs["dimension"][0]).size < 3 or np.asarray(recs["voxel_size"][0]).size < 3:
        raise NzError("invalid format")
    model.dim = tuple(int(v) for v in np.asarray(recs["dimension"][0]).reshape(-1)[:3])
    model.voxel_size = tuple(float(v) for v in np.asarray(recs["voxel_size"][0]).reshape(-1)[:3])
    for key in ("fov_strategy", "preproc", "orientation", "postproc"):
        if key in recs:
            setattr(model, key, _text(recs[key]))
    if "single_component_label" in recs:
        model.single_component_label = [int(v) for v in np.asarray(recs["single_component_label"][0]).reshape(-1, order="F")]
    model.testing_errors = [float(v) for v in np.asarray(recs.get("testing_errors", (np.zeros((3, 0), np.float32), 0))[0]).reshape(-1, order="F")]
    tr = [float(v) for v in np.asarray(recs.get("training_errors", (np.zeros((3, 0), np.float32), 0))[0]).reshape(-1, order="F")]
    model.training_errors = (tr + [0.0] * len(model.testing_errors))[: len(model.testing_errors)]    # main.cpp:188 resize
    model.train()
    arrays = []
    for i, p in enumerate(model.parameters()):
        key = "tensor%d" % i
        if key + ".slope" in recs or key + ".inter" in recs or (key in recs and recs[key][0].dtype.kind != "f"):
            raise NzError("%s is stored in TIPL's sloped (quantised) encoding (main.cpp:223-229), whose layout is defined by TIPL and not "
                          "available here: re-save the network with plain float tensors" % key)
        if key not in recs or recs[key][0].size != p.numel():
            got = recs[key][0].size if key in recs else 0
            raise NzError("tensor size mismatch at %s %d not the expected of size %d" % (key, got, p.numel()))
        arrays.append(np.asarray(recs[key][0], np.float32).reshape(-1, order="F"))
    model.load_parameters(arrays)
    return model
