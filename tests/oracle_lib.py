"""ctypes binding of oracle/libkzoracle.so — the CPU checker.  Test infrastructure only:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from kzero_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
GOLDEN = os.path.join(REPO, "tests", "golden")

_lib = None

TRACE_FN = C.CFUNCTYPE(None, C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p)


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "libkzoracle.so")
        src = os.path.join(ORACLE_DIR, "kz_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        L = C.CDLL(so)
        L.kzo_last_error.restype = C.c_char_p
        L.kzo_load.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.kzo_free.argtypes = [C.c_void_p]
        L.kzo_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.kzo_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.kzo_forward_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, TRACE_FN, C.c_void_p]
        L.kzo_encode_input_full.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p]
        L.kzo_encode_input_full.restype = None
        L.kzo_softmax_in_place.argtypes = [C.c_void_p, C.c_int]
        L.kzo_decode_output.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
        L.kzo_bits_push.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int]
        L.kzo_bits_push_block.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_uint64]
        L.kzo_bits_get.argtypes = [C.c_void_p, C.c_size_t]
        L.kzo_bits_storage_len.argtypes = [C.c_size_t]
        L.kzo_bits_storage_len.restype = C.c_size_t
        _lib = L
    return _lib


def _err():
    return lib().kzo_last_error().decode()


class OracleNet:
    def __init__(self, blob: bytes):
        self._h = C.c_void_p()
        self._blob = blob
        if lib().kzo_load(blob, len(blob), C.byref(self._h)) != 0:
            raise RuntimeError(_err())
        info = (C.c_int * 8)()
        lib().kzo_info(self._h, info)
        (self.c_in, self.h, self.w, self.n_scalar, self.n_bool, self.policy_len, self.depth, self.channels) = info

    def __del__(self):
        if getattr(self, "_h", None):
            lib().kzo_free(self._h)
            self._h = None

    def forward(self, x: np.ndarray, threads: int = 1):
        x = np.ascontiguousarray(x, dtype=np.float32)
        b = x.shape[0]
        assert x.shape[1:] == (self.c_in, self.h, self.w), x.shape
        scalars = np.empty((b, 5), np.float32)
        policy = np.empty((b, self.policy_len), np.float32)
        if lib().kzo_forward(self._h, x.ctypes.data, b, scalars.ctypes.data, policy.ctypes.data, threads) != 0:
            raise RuntimeError(_err())
        return scalars, policy

    def forward_trace(self, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float32)
        b = x.shape[0]
        scalars = np.empty((b, 5), np.float32)
        policy = np.empty((b, self.policy_len), np.float32)
        acts = {}

        def cb(name, data, count, board, _user):
            acts.setdefault(name.decode(), {})[board] = np.ctypeslib.as_array(data, shape=(count,)).copy()

        fn = TRACE_FN(cb)
        if lib().kzo_forward_trace(self._h, x.ctypes.data, b, scalars.ctypes.data, policy.ctypes.data, fn, None) != 0:
            raise RuntimeError(_err())
        stacked = {k: np.stack([v[i] for i in range(b)]) for k, v in acts.items()}
        return scalars, policy, stacked


def encode_input_full(bits: np.ndarray, scalars: np.ndarray, n_scalar, n_bool, h, w):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    scalars = np.ascontiguousarray(scalars, dtype=np.float32)
    b = bits.shape[0]
    out = np.empty((b, n_scalar + n_bool, h, w), np.float32)
    lib().kzo_encode_input_full(bits.ctypes.data, bits.shape[1], scalars.ctypes.data, b, n_scalar, n_bool, h, w,
                                out.ctypes.data)
    return out


def decode_output(scalars, logits, move_lists):
    scalars = np.ascontiguousarray(scalars, dtype=np.float32)
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    b, p = logits.shape
    offsets = np.zeros(b + 1, np.int64)
    offsets[1:] = np.cumsum([len(m) for m in move_lists])
    idx = np.concatenate([np.asarray(m, np.int32) for m in move_lists] + [np.zeros(0, np.int32)]).astype(np.int32)
    values = np.empty((b, 5), np.float32)
    pol = np.empty(max(len(idx), 1), np.float32)
    if lib().kzo_decode_output(scalars.ctypes.data, logits.ctypes.data, b, p, offsets.ctypes.data, idx.ctypes.data,
                               values.ctypes.data, pol.ctypes.data) != 0:
        raise RuntimeError(_err())
    return values, [pol[offsets[i]:offsets[i + 1]].copy() for i in range(b)]


def read_io(name, kind, c_in, h, w, policy_len):
    """Reads the reference's check format (python/lib/save_onnx.py:95-102)."""
    raw = open(os.path.join(GOLDEN, f"{name}.{kind}.io.bin"), "rb").read()
    b = raw[0]
    a = np.frombuffer(raw, np.float32, offset=1) if (len(raw) - 1) % 4 == 0 else None
    if a is None:
        a = np.frombuffer(raw[1:], np.float32)
    n_in = b * c_in * h * w
    x = a[:n_in].reshape(b, c_in, h, w)
    scalars = a[n_in:n_in + b * 5].reshape(b, 5)
    policy = a[n_in + b * 5:].reshape(b, policy_len)
    return x, scalars, policy


def read_packed(name, n_bool, n_scalar, h, w):
    raw = open(os.path.join(GOLDEN, f"{name}.planes.packed.bin"), "rb").read()
    b = raw[0]
    nbytes = (n_bool * h * w + 7) // 8
    bits = np.frombuffer(raw, np.uint8, count=b * nbytes, offset=1).reshape(b, nbytes)
    scalars = np.frombuffer(raw[1 + b * nbytes:], np.float32).reshape(b, n_scalar)
    return bits, scalars


GOLDEN_NETS = ["ataxx7_2x16", "ataxx7_4x64", "chess_2x32_att", "chess_2x32_dense_h", "chess_1x32_dense",
               "go9_2x16_conv", "go9_2x16_conv_terr",
               # round 5: the other games the server dispatches (server.rs:114-185)
               "arimaa_2x32", "ttt_2x16_dense", "sttt_2x16_dense_h",
               # PredictionHeads(AttentionTower, ...) (python/lib/model/attention.py; supervised_main_alpha.py:69-77)
               "chess_att2x64", "ataxx7_att2x32", "chess_att3x256",
               # DenseNetwork (python/lib/model/simple.py; the reference's own test networks, write_test_networks.py:14-18)
               "sttt_dn1x64", "sttt_dn1x64_res", "chess_dn3x96_res"]


def load_blob(name):
    return open(os.path.join(GOLDEN, f"{name}.kzm"), "rb").read()
