"""ctypes binding of libunet_hip.so (include/unet_hip.h).

This is the only way Python reaches the compute path: there is no fallback.  If the shared library is
missing or fails to load, importing this module raises -- the product path must fail loudly rather
than run anything else (the oracle is test infrastructure and is never imported from here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# UNET_HIP_LIBRARY: another build of the SAME library (csrc/tools/sanitize_host.sh: host code under ASan/UBSan); never a fallback
LIB_PATH = os.environ.get("UNET_HIP_LIBRARY") or os.path.join(_HERE, "libunet_hip.so")

DTYPE_F32, DTYPE_BF16 = 0, 1
IMPL_AUTO, IMPL_DIRECT = 0, 1

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libunet_hip.so not found at %s: build it first (python -c 'import __graft_entry__ as g; g.build()' "
        "or unet-studio_amd/csrc/build.sh); there is no CPU fallback" % LIB_PATH)

# Load order matters: libunet_hip.so needs libamdhip64.so.7.  PyTorch-ROCm ships its own copy (torch/lib); when torch is loaded
# first the dynamic linker resolves our dependency to that already-loaded runtime and the process has ONE HIP runtime that owns
# both torch's allocations and our launches.  Loaded the other way round, /opt/rocm's copy comes in for us, torch brings its own,
# and the second runtime finds no device ("no ROCm-capable device is detected", profiles/hip_runtime_load_order.py).  torch is the
# allocator / stream provider of every caller of this module, so it is imported here, before the library.
import torch  # noqa: E402,F401

lib = C.CDLL(LIB_PATH)

_vp, _i, _sz, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_float
_pp = C.POINTER(C.c_void_p)


def _sig(name, res, *args):
    fn = getattr(lib, name)
    fn.restype = res
    fn.argtypes = list(args)
    return fn


_sig("unet_last_error", C.c_char_p)
_sig("unet_init", _i, C.POINTER(_i))
_sig("unet_device_info", _i, _i, C.c_char_p, _sz, C.POINTER(_sz), C.POINTER(_i), C.POINTER(_i))
_sig("unet_plan_create", _i, C.c_char_p, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp))
_sig("unet_plan_destroy", None, _vp)
_sig("unet_plan_param_count", _i, _vp, C.POINTER(_i))
_sig("unet_plan_param_shape", _i, _vp, _i, C.POINTER(C.c_int64), C.POINTER(_i))
_sig("unet_plan_param_name", _i, _vp, _i, C.c_char_p, _sz)
_sig("unet_plan_param_decay", _i, _vp, _i, C.POINTER(_i))
_sig("unet_plan_param_fan_in", _i, _vp, _i, C.POINTER(C.c_int64), C.POINTER(_i))
_sig("unet_plan_buffer_count", _i, _vp, C.POINTER(_i))
_sig("unet_plan_buffer_shape", _i, _vp, _i, C.POINTER(C.c_int64))
_sig("unet_plan_output_count", _i, _vp, C.POINTER(_i))
_sig("unet_plan_output_shape", _i, _vp, _i, C.POINTER(C.c_int64))
_sig("unet_plan_workspace_bytes", _i, _vp, C.POINTER(_sz))
_sig("unet_plan_flops", _i, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double))
_sig("unet_plan_describe", _sz, _vp, C.c_char_p, _sz)
_sig("unet_plan_op_count", _i, _vp, C.POINTER(_i))
_sig("unet_plan_op_info", _i, _vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_int64),
     C.POINTER(C.c_int64), C.c_char_p, _sz)
_sig("unet_profile_begin", _i)
_sig("unet_profile_end", _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), C.POINTER(_i))
_sig("unet_forward", _i, _vp, _pp, _pp, _vp, _pp, _vp, _i, _vp)
_sig("unet_backward", _i, _vp, _pp, _pp, _pp, _vp, _vp, _vp)
_sig("unet_backward_part", _i, _vp, _pp, _pp, _pp, _vp, _vp, _i, _i, _vp)
_sig("unet_plan_backward_buckets", _i, _vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_int64))
_sig("unet_loss_scratch_bytes", _i, _vp, C.POINTER(_sz))
_sig("unet_loss", _i, _vp, _pp, _vp, _i, _i, _pp, _vp, _vp, _vp)
_sig("unet_forward_loss", _i, _vp, _pp, _pp, _vp, _pp, _vp, _i, _i, _pp, _vp, _vp, _vp, _vp)
_sig("unet_forward_loss_mode", _i, _vp, _pp, _pp, _vp, _pp, _vp, _i, _i, _pp, _vp, _vp, _vp, _i, _vp)
_sig("unet_sum_buffers", _i, _pp, _i, _vp, C.c_int64, _i, _vp)
_sig("unet_stream_create_cu_range", _i, _i, _i, _i, _vp)
_sig("unet_stream_destroy", _i, _vp)
_sig("unet_plan_side_cu_range", _i, _vp, _i, _i)
_sig("unet_pack_filters", _i, _vp, _vp, _vp, _i, _vp, _vp)
_sig("unet_sgd_step", _i, _vp, _vp, _vp, _vp, _f, _f, _i, _f, _f, _f, _vp, _vp, _vp)
_sig("unet_set_error", None, C.c_char_p)
_sig("unet_comm_unique_id", _i, _vp)
_sig("unet_comm_create", _i, _i, _i, _vp, _i, C.POINTER(_vp))
_sig("unet_comm_create_all", _i, _i, C.POINTER(_i), C.POINTER(_vp))
_sig("unet_comm_destroy", _i, _vp)
_sig("unet_comm_rank", _i, _vp, C.POINTER(_i), C.POINTER(_i))
_sig("unet_allreduce_grads", _i, _vp, _vp, C.c_int64, C.c_int64, _vp)
_sig("unet_allreduce_grads_all", _i, _pp, _i, _pp, C.c_int64, C.c_int64, _pp)
_sig("unet_comm_broadcast", _i, _vp, _vp, C.c_int64, _i, _vp)
_sig("unet_comm_join", _i, _vp, _vp)
_sig("unet_op_scratch_bytes", _i, _i, _i, _i, _i, _i, C.POINTER(_sz))
_sig("unet_op_conv3d_fwd", _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp)
_sig("unet_op_conv3d_pack", _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp)
_sig("unet_op_conv3d_fwd_packed", _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp)
_sig("unet_op_conv3d_fwd_fused", _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp)
_sig("unet_op_conv3d_bwd_data", _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp)
_sig("unet_op_conv3d_bwd_weight", _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp)
_sig("unet_op_convt_fwd", _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp)
_sig("unet_op_convt_bwd_data", _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp)
_sig("unet_op_convt_bwd_weight", _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp)
_sig("unet_op_pack_ndhwc", _i, _i, _vp, _vp, _i, C.c_int64, _vp)
_sig("unet_op_unpack_ncdhw", _i, _i, _vp, _vp, _i, C.c_int64, _vp)

# every symbol include/unet_hip.h declares (tests check that the library exports all of them)
EXPORTS = [
    "unet_last_error", "unet_init", "unet_device_info", "unet_plan_create", "unet_plan_destroy", "unet_plan_param_count",
    "unet_plan_param_shape", "unet_plan_param_name", "unet_plan_param_decay", "unet_plan_param_fan_in", "unet_plan_buffer_count",
    "unet_plan_buffer_shape", "unet_plan_output_count", "unet_plan_output_shape", "unet_plan_workspace_bytes",
    "unet_plan_flops", "unet_plan_describe", "unet_plan_op_count", "unet_plan_op_info", "unet_profile_begin", "unet_profile_end", "unet_forward", "unet_backward", "unet_backward_part", "unet_plan_backward_buckets", "unet_loss_scratch_bytes", "unet_loss", "unet_forward_loss", "unet_forward_loss_mode", "unet_sum_buffers",
    "unet_sgd_step", "unet_pack_filters", "unet_stream_create_cu_range", "unet_stream_destroy", "unet_plan_side_cu_range", "unet_set_error", "unet_comm_unique_id", "unet_comm_create", "unet_comm_create_all", "unet_comm_destroy", "unet_comm_rank",
    "unet_allreduce_grads", "unet_allreduce_grads_all", "unet_comm_broadcast", "unet_comm_join", "unet_op_scratch_bytes", "unet_op_conv3d_fwd", "unet_op_conv3d_fwd_fused", "unet_op_conv3d_pack", "unet_op_conv3d_fwd_packed", "unet_op_conv3d_bwd_data", "unet_op_conv3d_bwd_weight",
    "unet_op_convt_fwd", "unet_op_convt_bwd_data", "unet_op_convt_bwd_weight", "unet_op_pack_ndhwc", "unet_op_unpack_ncdhw",
]


MODE_PACKS_CURRENT = 2   # include/unet_hip.h UNET_MODE_PACKS_CURRENT


class UNetError(RuntimeError):
    """What the C++ host rethrows as std::runtime_error (unet.cpp:53,66,88,117)."""


def check(rc):
    if rc != 0:
        raise UNetError(lib.unet_last_error().decode("utf-8", "replace"))


def ptr_array(ptrs):
    """host array of device pointers (None -> NULL)"""
    arr = (C.c_void_p * max(1, len(ptrs)))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


class Plan:
    """unet_plan: architecture DSL bound to one input size, element type and device."""

    def __init__(self, arch, in_c, out_c, size, dtype=DTYPE_BF16, device=0, impl=IMPL_AUTO):
        self.handle = C.c_void_p()
        self._pid = os.getpid()
        D, H, W = size
        check(lib.unet_plan_create(arch.encode(), in_c, out_c, D, H, W, dtype, device, impl, C.byref(self.handle)))
        self.arch, self.in_c, self.out_c, self.size, self.dtype, self.device = arch, in_c, out_c, (D, H, W), dtype, device
        n = C.c_int()
        check(lib.unet_plan_param_count(self.handle, C.byref(n)))
        self.param_shapes, self.param_decay, self.param_fan_in, self.param_is_norm_weight, self.param_names = [], [], [], [], []
        dims, nd, dec, fan, isn = (C.c_int64 * 5)(), C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        for i in range(n.value):
            check(lib.unet_plan_param_shape(self.handle, i, dims, C.byref(nd)))
            self.param_shapes.append(tuple(dims[k] for k in range(nd.value)))
            nm = C.create_string_buffer(128)
            check(lib.unet_plan_param_name(self.handle, i, nm, 128))
            self.param_names.append(nm.value.decode())
            check(lib.unet_plan_param_decay(self.handle, i, C.byref(dec)))
            self.param_decay.append(bool(dec.value))
            check(lib.unet_plan_param_fan_in(self.handle, i, C.byref(fan), C.byref(isn)))
            self.param_fan_in.append(fan.value)
            self.param_is_norm_weight.append(bool(isn.value))
        check(lib.unet_plan_buffer_count(self.handle, C.byref(n)))
        self.buffer_numel = []
        ne = C.c_int64()
        for i in range(n.value):
            check(lib.unet_plan_buffer_shape(self.handle, i, C.byref(ne)))
            self.buffer_numel.append(ne.value)
        check(lib.unet_plan_output_count(self.handle, C.byref(n)))
        self.output_shapes = []
        for l in range(n.value):
            check(lib.unet_plan_output_shape(self.handle, l, dims))
            self.output_shapes.append(tuple(dims[k] for k in range(5)))
        b = C.c_size_t()
        check(lib.unet_plan_workspace_bytes(self.handle, C.byref(b)))
        self.workspace_bytes = b.value
        check(lib.unet_loss_scratch_bytes(self.handle, C.byref(b)))
        self.loss_scratch_bytes = b.value
        f, g = C.c_double(), C.c_double()
        check(lib.unet_plan_flops(self.handle, C.byref(f), C.byref(g)))
        self.flops_fwd, self.flops_bwd = f.value, g.value

    def backward_buckets(self, max_buckets=3):
        """[(op_lo, elem_lo)]: running the backward for ops [op_lo[k], op_lo[k-1]) finishes the gradients of the flat parameter
        elements [elem_lo[k], elem_lo[k-1]) -- the unit an overlapped all-reduce works on"""
        n = C.c_int()
        ops, el = (C.c_int * max_buckets)(), (C.c_int64 * max_buckets)()
        check(lib.unet_plan_backward_buckets(self.handle, max_buckets, C.byref(n), ops, el))
        return [(ops[k], el[k]) for k in range(n.value)]

    def ops(self):
        """lowered op list: dicts with kind, cin, cout, ks, stride, in_dims, out_dims, name (unet_plan_op_info)"""
        n = C.c_int()
        check(lib.unet_plan_op_count(self.handle, C.byref(n)))
        out = []
        k, ci, co, ks, st = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        di, do = (C.c_int64 * 3)(), (C.c_int64 * 3)()
        for i in range(n.value):
            nm = C.create_string_buffer(128)
            check(lib.unet_plan_op_info(self.handle, i, C.byref(k), C.byref(ci), C.byref(co), C.byref(ks), C.byref(st), di, do, nm, 128))
            out.append({"kind": k.value, "cin": ci.value, "cout": co.value, "ks": ks.value, "stride": st.value,
                        "in_dims": tuple(di), "out_dims": tuple(do), "name": nm.value.decode()})
        return out

    def describe(self):
        n = lib.unet_plan_describe(self.handle, None, 0)
        buf = C.create_string_buffer(n)
        lib.unet_plan_describe(self.handle, buf, n)
        return buf.value.decode()

    def __del__(self):
        h = getattr(self, "handle", None)
        # `lib` is None once the interpreter tears the module down.  A forked child (multiprocessing.Manager, a DataLoader worker)
        # inherits the object but must not touch the parent's GPU state: HIP is not fork-safe, and a garbage collection in such a
        # child used to end in hipFree / hipStreamDestroy there (segmentation fault in the Manager process of the GPU dist tests).
        if h is not None and h.value and lib is not None and getattr(self, "_pid", None) == os.getpid():
            lib.unet_plan_destroy(h)
            self.handle = C.c_void_p()


PROF_CATEGORIES = ["conv_fwd", "dgrad", "wgrad", "norm_fwd", "norm_bwd", "other"]


class profile:
    """with engine.profile() as pr: ...forward/backward calls of this thread...; pr.records = [(op_index, category, ms)]"""

    def __init__(self, max_records=1 << 16):
        self.max_records, self.records = max_records, []

    def __enter__(self):
        check(lib.unet_profile_begin())
        return self

    def __exit__(self, *exc):
        n = C.c_int()
        op, cat, ms = (C.c_int * self.max_records)(), (C.c_int * self.max_records)(), (C.c_float * self.max_records)()
        rc = lib.unet_profile_end(self.max_records, op, cat, ms, C.byref(n))
        self.records = [(op[i], PROF_CATEGORIES[cat[i]], ms[i]) for i in range(n.value)]
        if exc[0] is None:
            check(rc)
        return False


COMM_ID_BYTES = 128


class Comm:
    """unet_comm: RCCL communicator under the C ABI (one process per GPU).  The 128-byte id is made on rank 0 and handed to the
    other ranks out of band; `from_torch_distributed` uses an existing process group (any backend) as that side channel only."""

    def __init__(self, rank, world, id_bytes, device=0):
        self.handle = C.c_void_p()
        self._pid = os.getpid()
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        check(lib.unet_comm_create(rank, world, buf, device, C.byref(self.handle)))
        self.rank, self.world, self.device = rank, world, device

    @staticmethod
    def unique_id():
        buf = (C.c_char * COMM_ID_BYTES)()
        check(lib.unet_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def single(cls, device=0):
        return cls(0, 1, cls.unique_id(), device)

    @classmethod
    def from_torch_distributed(cls, device, group=None):
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(rank, world, box[0], device)

    def allreduce(self, flat, lo, hi, stream):
        """flat[lo:hi] (fp32, device) <- sum over ranks; enqueued after `stream`'s work so far, on the communicator's own stream"""
        check(lib.unet_allreduce_grads(self.handle, flat.data_ptr(), int(lo), int(hi), stream))

    def broadcast(self, buf, root, stream):
        check(lib.unet_comm_broadcast(self.handle, buf.data_ptr(), int(buf.numel()), int(root), stream))

    def join(self, stream):
        check(lib.unet_comm_join(self.handle, stream))

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value and lib is not None and getattr(self, "_pid", None) == os.getpid():   # never from a forked child
            lib.unet_comm_destroy(h)
            self.handle = C.c_void_p()
