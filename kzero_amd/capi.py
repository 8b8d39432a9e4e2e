"""ctypes binding of the C ABI in include/kz_hip.h (libkzhip.so).

This is exactly what a foreign host binds; the Rust equivalent is shown in INTEGRATION.md.  There is no fallback:
if the library is missing or a call fails, this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KZ_LIB_PATH") or os.path.join(_HERE, "libkzhip.so")

KZ_DTYPE_F32 = 0
KZ_DTYPE_F16 = 1
KZ_DTYPE_F32_SPLIT16 = 2
KZ_ENGINE_SLOTS = 4

POLICY_KINDS = {0: "ataxx_conv", 1: "conv", 2: "attention", 3: "dense"}


class KzError(RuntimeError):
    pass


class PathPlan(C.Structure):
    _fields_ = [("tower_path", C.c_char * 48), ("launches_per_batch", C.c_int32), ("reserved", C.c_int32 * 3)]


class ModelInfo(C.Structure):
    _fields_ = [
        ("input_channels", C.c_int32), ("board_h", C.c_int32), ("board_w", C.c_int32),
        ("input_scalar_channels", C.c_int32), ("input_bool_channels", C.c_int32), ("policy_len", C.c_int32),
        ("tower_depth", C.c_int32), ("tower_channels", C.c_int32), ("policy_kind", C.c_int32),
        ("bits_bytes", C.c_int32), ("param_count", C.c_int64), ("flops_per_eval", C.c_double),
    ]


# name -> (restype, argtypes); every symbol include/kz_hip.h declares
SIGNATURES = {
    "kz_last_error": (C.c_char_p, []),
    "kz_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "kz_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "kz_model_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "kz_model_load_memory": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "kz_model_load_onnx": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "kz_model_load_onnx_memory": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "kz_model_free": (None, [C.c_void_p]),
    "kz_model_get_info": (C.c_int, [C.c_void_p, C.POINTER(ModelInfo)]),
    "kz_engine_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "kz_engine_destroy": (None, [C.c_void_p]),
    "kz_model_supports_dtype": (C.c_int, [C.c_void_p, C.c_int]),
    "kz_model_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "kz_engine_max_batch": (C.c_int, [C.c_void_p]),
    "kz_engine_eval_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "kz_engine_eval_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "kz_engine_eval_packed_decoded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "kz_engine_submit_packed": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]),
    "kz_engine_wait": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "kz_engine_wait_view": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "kz_engine_submit_packed_decoded": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
                                                  C.c_void_p, C.c_void_p]),
    "kz_engine_wait_decoded": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "kz_engine_enqueue_packed_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
                                                  C.c_void_p, C.c_void_p]),
    "kz_engine_enqueue_dense_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "kz_engine_synchronize": (C.c_int, [C.c_void_p]),
    "kz_device_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "kz_device_free": (C.c_int, [C.c_int, C.c_void_p]),
    "kz_memcpy_h2d": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "kz_memcpy_d2h": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "kz_device_synchronize": (C.c_int, [C.c_int]),
    "kz_engine_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "kz_engine_kernel_time": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "kz_engine_tower_path": (C.c_char_p, [C.c_void_p]),
    "kz_engine_launch_geometry": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kz_engine_read_activation": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p]),
}

_lib = None


def load():
    """Loads libkzhip.so.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KzError(f"{LIB_PATH} is missing: build it with kzero_amd/csrc/build.sh "
                          f"(or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            if "KZ_LIB_PATH" in os.environ and not hasattr(lib, name):
                continue  # (an older build named explicitly for a same-box A/B: its missing entry points fail when called)
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise KzError(load().kz_last_error().decode())


def device_count() -> int:
    n = C.c_int()
    check(load().kz_device_count(C.byref(n)))
    return n.value


def device_pci_bus_id(device: int) -> str:
    buf = C.create_string_buffer(64)
    check(load().kz_device_pci_bus_id(device, buf, len(buf)))
    return buf.value.decode()


class Model:
    """`Arc<Graph>`: immutable, shareable across engines, threads and devices."""

    def __init__(self, blob: bytes = None, path: str = None, onnx_scalar_channels: int = None):
        """blob/path: a KZMODEL1 container or an ONNX file; onnx_scalar_channels: for ONNX, how many input planes are
        broadcast scalars (the mapper's input_scalar_count) — needed for the packed-input entry points."""
        self._h = C.c_void_p()
        if onnx_scalar_channels is not None:
            if blob is not None:
                check(load().kz_model_load_onnx_memory(blob, len(blob), onnx_scalar_channels, C.byref(self._h)))
            else:
                check(load().kz_model_load_onnx(path.encode(), onnx_scalar_channels, C.byref(self._h)))
        elif blob is not None:
            check(load().kz_model_load_memory(blob, len(blob), C.byref(self._h)))
        else:
            check(load().kz_model_load(path.encode(), C.byref(self._h)))
        info = ModelInfo()
        check(load().kz_model_get_info(self._h, C.byref(info)))
        self.info = info

    def plan(self, max_batch: int, dtype: int):
        """(tower path, launches per batch) kz_engine_create would choose — host logic, no GPU needed."""
        out = PathPlan()
        check(load().kz_model_plan(self._h, max_batch, dtype, C.byref(out)))
        return out.tower_path.decode(), out.launches_per_batch

    def supports_dtype(self, dtype: int) -> bool:
        return load().kz_model_supports_dtype(self._h, dtype) == 1

    def close(self):
        if self._h:
            load().kz_model_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    def __init__(self, device: int, nbytes: int):
        self.device, self.nbytes = device, nbytes
        self.ptr = C.c_void_p()
        check(load().kz_device_malloc(device, nbytes, C.byref(self.ptr)))

    @classmethod
    def from_host(cls, device: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        buf = cls(device, arr.nbytes)
        if arr.nbytes:
            check(load().kz_memcpy_h2d(device, buf.ptr, arr.ctypes.data, arr.nbytes))
        return buf

    def to_host(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            check(load().kz_memcpy_d2h(self.device, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            load().kz_device_free(self.device, self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One per executor thread, like `CudaNetwork` (rust/kz-core/src/network/cudnn.rs:18-43)."""

    def __init__(self, model: Model, device: int, max_batch: int, dtype: int):
        self.model, self.device, self.dtype = model, device, dtype
        self._h = C.c_void_p()
        check(load().kz_engine_create(model._h, device, max_batch, dtype, C.byref(self._h)))
        self.max_batch = load().kz_engine_max_batch(self._h)

    @property
    def tower_path(self) -> str:
        return load().kz_engine_tower_path(self._h).decode()

    def launch_geometry(self, batch: int):
        """(workgroups per launch, boards per workgroup) of the path's dominant launch at `batch`."""
        wgs, per = C.c_int(), C.c_int()
        check(load().kz_engine_launch_geometry(self._h, batch, C.byref(wgs), C.byref(per)))
        return wgs.value, per.value

    def eval_dense(self, x: np.ndarray):
        info = self.model.info
        x = np.ascontiguousarray(x, dtype=np.float32)
        batch = x.shape[0]
        scalars = np.empty((batch, 5), np.float32)
        policy = np.empty((batch, info.policy_len), np.float32)
        check(load().kz_engine_eval_dense(self._h, x.ctypes.data, batch, scalars.ctypes.data, policy.ctypes.data))
        return scalars, policy

    def eval_packed(self, bits: np.ndarray, scalars_in: np.ndarray):
        info = self.model.info
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        scalars_in = np.ascontiguousarray(scalars_in, dtype=np.float32)
        batch = bits.shape[0]
        scalars = np.empty((batch, 5), np.float32)
        policy = np.empty((batch, info.policy_len), np.float32)
        stride = bits.shape[1] if bits.ndim == 2 else 0
        check(load().kz_engine_eval_packed(self._h, bits.ctypes.data, stride, scalars_in.ctypes.data, batch,
                                           scalars.ctypes.data, policy.ctypes.data))
        return scalars, policy

    def eval_packed_decoded(self, bits: np.ndarray, scalars_in: np.ndarray, move_lists):
        """decode_output on the GPU: returns (values [batch,5], [probs per board])."""
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        scalars_in = np.ascontiguousarray(scalars_in, dtype=np.float32)
        batch = bits.shape[0]
        offsets = np.zeros(batch + 1, np.int64)
        offsets[1:] = np.cumsum([len(m) for m in move_lists])
        idx = np.ascontiguousarray(np.concatenate([np.asarray(m, np.int32) for m in move_lists] + [np.zeros(0, np.int32)]),
                                   dtype=np.int32)
        values = np.empty((batch, 5), np.float32)
        probs = np.empty(max(len(idx), 1), np.float32)
        check(load().kz_engine_eval_packed_decoded(self._h, bits.ctypes.data, bits.shape[1] if bits.ndim == 2 else 0,
                                                   scalars_in.ctypes.data, batch, offsets.ctypes.data, idx.ctypes.data,
                                                   values.ctypes.data, probs.ctypes.data))
        return values, [probs[offsets[i]:offsets[i + 1]].copy() for i in range(batch)]

    def submit_packed(self, slot: int, bits: np.ndarray, scalars_in: np.ndarray):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        scalars_in = np.ascontiguousarray(scalars_in, dtype=np.float32)
        check(load().kz_engine_submit_packed(self._h, slot, bits.ctypes.data, bits.shape[1], scalars_in.ctypes.data,
                                             bits.shape[0]))
        return bits.shape[0]

    def wait(self, slot: int, batch: int):
        scalars = np.empty((batch, 5), np.float32)
        policy = np.empty((batch, self.model.info.policy_len), np.float32)
        check(load().kz_engine_wait(self._h, slot, scalars.ctypes.data, policy.ctypes.data))
        return scalars, policy

    def submit_packed_decoded(self, slot: int, bits: np.ndarray, scalars_in: np.ndarray, move_lists):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        scalars_in = np.ascontiguousarray(scalars_in, dtype=np.float32)
        offsets = np.zeros(len(move_lists) + 1, np.int64)
        offsets[1:] = np.cumsum([len(m) for m in move_lists])
        idx = np.ascontiguousarray(np.concatenate([np.asarray(m, np.int32) for m in move_lists]) if offsets[-1] else
                                   np.zeros(0, np.int32))
        check(load().kz_engine_submit_packed_decoded(self._h, slot, bits.ctypes.data, bits.shape[1],
                                                     scalars_in.ctypes.data, bits.shape[0], offsets.ctypes.data,
                                                     idx.ctypes.data))
        return offsets

    def submit_packed_decoded_csr(self, slot: int, bits: np.ndarray, scalars_in: np.ndarray, offsets: np.ndarray, idx: np.ndarray):
        """The same with the CSR move lists already built (contiguous uint8 / float32 / int64 / int32 arrays): what a timed
        loop calls."""
        check(load().kz_engine_submit_packed_decoded(self._h, slot, bits.ctypes.data, bits.shape[1], scalars_in.ctypes.data,
                                                     bits.shape[0], offsets.ctypes.data, idx.ctypes.data))

    def wait_decoded_view(self, slot: int):
        """kz_engine_wait_decoded without copies: the two pointers into the slot's pinned staging."""
        pv, pp = C.c_void_p(), C.c_void_p()
        check(load().kz_engine_wait_decoded(self._h, slot, C.byref(pv), C.byref(pp)))
        return pv.value, pp.value

    def wait_decoded(self, slot: int, offsets: np.ndarray):
        """values [batch,5] and one probability array per board: copies of the slot's pinned staging."""
        pv, pp = C.c_void_p(), C.c_void_p()
        check(load().kz_engine_wait_decoded(self._h, slot, C.byref(pv), C.byref(pp)))
        batch, total = len(offsets) - 1, int(offsets[-1])
        values = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_float)), shape=(batch, 5)).copy() if batch else np.empty((0, 5), np.float32)
        probs = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_float)), shape=(total,)).copy() if total else np.zeros(0, np.float32)
        return values, [probs[offsets[i]:offsets[i + 1]] for i in range(batch)]

    def wait_view(self, slot: int, batch: int):
        """Zero-copy wait: arrays over the slot's pinned staging, valid until the next submit on that slot."""
        ps, pp = C.c_void_p(), C.c_void_p()
        check(load().kz_engine_wait_view(self._h, slot, C.byref(ps), C.byref(pp)))
        if batch == 0:
            return np.empty((0, 5), np.float32), np.empty((0, self.model.info.policy_len), np.float32)
        plen = self.model.info.policy_len
        scalars = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_float)), shape=(batch, 5))
        policy = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_float)), shape=(batch, plen))
        return scalars, policy

    def enqueue_packed_device(self, d_bits: DeviceBuffer, stride: int, d_scalars: DeviceBuffer, batch: int,
                              d_scalars_out: DeviceBuffer, d_policy_out: DeviceBuffer):
        check(load().kz_engine_enqueue_packed_device(self._h, d_bits.ptr, stride, d_scalars.ptr, batch,
                                                     d_scalars_out.ptr, d_policy_out.ptr))

    def enqueue_dense_device(self, d_input: DeviceBuffer, batch: int, d_scalars_out: DeviceBuffer,
                             d_policy_out: DeviceBuffer):
        check(load().kz_engine_enqueue_dense_device(self._h, d_input.ptr, batch, d_scalars_out.ptr, d_policy_out.ptr))

    def synchronize(self):
        check(load().kz_engine_synchronize(self._h))

    def set_profiling(self, on: bool):
        check(load().kz_engine_set_profiling(self._h, int(on)))

    def kernel_time(self, prefix: str):
        total, n = C.c_double(), C.c_int64()
        check(load().kz_engine_kernel_time(self._h, prefix.encode(), C.byref(total), C.byref(n)))
        return total.value, n.value

    def read_activation(self, name: str, batch: int) -> np.ndarray:
        info = self.model.info
        out = np.empty((batch, info.tower_channels, info.board_h, info.board_w), np.float32)
        check(load().kz_engine_read_activation(self._h, name.encode(), batch, out.ctypes.data))
        return out

    def close(self):
        if self._h:
            load().kz_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
