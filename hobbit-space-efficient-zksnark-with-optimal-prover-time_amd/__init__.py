"""hobbit_amd -- Python harness over libhobbit_hip.so (the C ABI in include/hobbit_hip.h).

This module is plumbing for tests and bench.py: it loads the in-tree shared library, owns device
buffers through the ABI's own allocator, and mirrors the reference's function names for the hot
path (commit_standard, generate_2product_sumcheck_proof, expander_init_store, ...; reference
src/Our_PC.cpp, src/sumcheck.cpp, src/expanders.h) so the parity tests read like calls into the
reference.  The product itself is the HIP library; the C++ host mirror with the reference's exact
C++ signatures lives in host/.

There is NO CPU fallback: if the library is missing, or no HIP device is present, construction
raises.  Nothing under oracle/ is imported here.
"""
import ctypes
import importlib.util
import os
import sys
import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HOBBIT_HIP_LIB") or os.path.join(PKG_DIR, "libhobbit_hip.so")      # (HOBBIT_HIP_LIB: A/B runs of two builds in one gpurun call)
P = (1 << 61) - 1

c_sz = ctypes.c_size_t
c_vp = ctypes.c_void_p
c_int = ctypes.c_int
c_ll = ctypes.c_longlong

# every symbol include/hobbit_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "hobbit_ctx_create", "hobbit_ctx_create_on_stream", "hobbit_ctx_destroy", "hobbit_last_error", "hobbit_version", "hobbit_sync",
    "hobbit_malloc", "hobbit_free", "hobbit_memcpy_h2d", "hobbit_memcpy_d2h", "hobbit_memset", "hobbit_timer_begin", "hobbit_timer_end_ms",
    "hobbit_profile_enable", "hobbit_profile_reset", "hobbit_profile_get", "hobbit_profile_names",
    "hobbit_mimc", "hobbit_f_mul_host", "hobbit_f_inv_host", "hobbit_f_binop",
    "hobbit_graph_reset", "hobbit_graph_upload", "hobbit_graph_finalize", "hobbit_encode_batch", "hobbit_fft_batch",
    "hobbit_blake3_64", "hobbit_hash_md", "hobbit_mt_commit_blake", "hobbit_merkle_levels", "hobbit_merkle_path", "hobbit_merkle_paths",
    "hobbit_eq_table", "hobbit_eval_vector", "hobbit_tensorcode",
    "hobbit_commit_standard", "hobbit_commit_standard_host", "hobbit_commitment_free", "hobbit_commitment_num_leaves", "hobbit_commitment_levels_dev",
    "hobbit_commitment_tensor_dev", "hobbit_commitment_levels", "hobbit_commitment_root", "hobbit_commitment_tensor_row",
    "hobbit_commitment_gather", "hobbit_commitment_path", "hobbit_commitment_paths",
    "hobbit_elastic_begin", "hobbit_elastic_push", "hobbit_elastic_push_inner", "hobbit_elastic_finish", "hobbit_elastic_free",
    "hobbit_elastic_open_begin", "hobbit_elastic_open_aggregate_push", "hobbit_elastic_open_aggregate_finish", "hobbit_elastic_open_reply_push",
    "hobbit_elastic_open_finish", "hobbit_elastic_open_free", "hobbit_elastic_open_begin_lin", "hobbit_elastic_open_dims", "hobbit_transcript_record", "hobbit_transcript_count", "hobbit_transcript_read", "hobbit_generate_randomness",
    "hobbit_read_mul_tree_layer", "hobbit_read_mul_tree_data", "hobbit_generate_claims_opt", "hobbit_sumcheck3_stream_batch", "hobbit_mul_tree_stream_shallow",
    "hobbit_gate_consistency_stream", "hobbit_set_lookups", "hobbit_gate_consistency_lookups_stream", "hobbit_open_standard_rs", "hobbit_leaf_chain_relay", "hobbit_verify_path_host", "hobbit_fingerprint_map", "hobbit_leaf_chain", "hobbit_axpy_aggregate", "hobbit_stream_fold",
    "hobbit_tensorcode_chunks", "hobbit_inner_digests", "hobbit_chain_digests", "hobbit_blake3_64_host",
    "hobbit_parity_matrix", "hobbit_phi_g", "hobbit_prepare_matrix_cols", "hobbit_prove_linear_code", "hobbit_prove_fft",
    "hobbit_prove_fft_matrix",
    "hobbit_whir_prove", "hobbit_shockwave_prove",
    "hobbit_batch_3product_sumcheck", "hobbit_mul_tree", "hobbit_shockwave_commit", "hobbit_change_form", "hobbit_whir_commit",
    "hobbit_open_core", "hobbit_open_standard", "hobbit_open_from_aggregate", "hobbit_tensor_gather", "hobbit_u64_bias_fold", "hobbit_gate_sumcheck", "hobbit_compute2p_error_terms", "hobbit_compute3p_error_terms", "hobbit_compute4p_error_terms", "hobbit_fold_axpy",
    "hobbit_fold_axpy_i32", "hobbit_batch_prod",
    "hobbit_aggregate", "hobbit_sumcheck2", "hobbit_sumcheck3", "hobbit_fill_splitmix",
]



class _OpenOut(ctypes.Structure):
    """hobbit_open_out (include/hobbit_hip.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("cols", "rows", "reply", "paths", "qpoly", "r", "vr", "fin", "scalars", "checks", "roots", "sp_c", "sp_f")]


class _ShockwaveOut(ctypes.Structure):
    """hobbit_shockwave_out (include/hobbit_hip.h); field order = Hobbit._SP_NAMES"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("I", "q1", "r1", "vr1", "fin1", "q2", "r2", "vr2", "fin2", "wq", "wa", "wroots", "wscal", "wchecks", "whir_root", "iters",
                                              "reply", "paths", "qidx", "qreply", "qpaths", "final_pb", "qn")]

class HobbitError(RuntimeError):
    pass


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise HobbitError("libhobbit_hip.so is not built (run __graft_entry__.build()): " + path)
    lib = ctypes.CDLL(path)
    lib.hobbit_last_error.restype = ctypes.c_char_p
    lib.hobbit_version.restype = ctypes.c_char_p
    lib.hobbit_commitment_num_leaves.restype = c_sz
    lib.hobbit_commitment_levels_dev.restype = c_vp
    lib.hobbit_commitment_tensor_dev.restype = c_vp
    # explicit prototypes: a bare Python int would otherwise be passed as a 32-bit C int and
    # truncate device pointers / sizes
    V, S, I, L, U64 = c_vp, c_sz, c_int, c_ll, ctypes.c_uint64
    protos = {
        "hobbit_ctx_create": [I, V], "hobbit_ctx_create_on_stream": [I, V, V], "hobbit_ctx_destroy": [V], "hobbit_last_error": [V],
        "hobbit_sync": [V], "hobbit_malloc": [V, S, V], "hobbit_free": [V, V], "hobbit_memcpy_h2d": [V, V, V, S],
        "hobbit_memcpy_d2h": [V, V, V, S], "hobbit_memset": [V, V, I, S], "hobbit_timer_begin": [V], "hobbit_timer_end_ms": [V, V],
        "hobbit_profile_enable": [V, I], "hobbit_profile_reset": [V], "hobbit_profile_get": [V, ctypes.c_char_p, V, V],
        "hobbit_profile_names": [V, V, S], "hobbit_mimc": [V, V, V], "hobbit_f_mul_host": [V, V, V, S], "hobbit_f_inv_host": [V, V, S],
        "hobbit_f_binop": [V, I, V, V, V, S], "hobbit_graph_reset": [V], "hobbit_graph_upload": [V, I, I, L, L, I, V, V],
        "hobbit_graph_finalize": [V, L, V], "hobbit_encode_batch": [V, V, V, L, S, S, S], "hobbit_fft_batch": [V, V, I, S, S, I],
        "hobbit_blake3_64": [V, V, V, S], "hobbit_hash_md": [V, V, V, V, S], "hobbit_mt_commit_blake": [V, V, S, V],
        "hobbit_merkle_levels": [V, V, S, I], "hobbit_merkle_path": [V, V, S, S, V], "hobbit_merkle_paths": [V, V, S, V, S, V],
        "hobbit_eq_table": [V, V, I, V], "hobbit_eval_vector": [V, V, S, V, V], "hobbit_tensorcode": [V, V, S, I, I, V],
        "hobbit_commit_standard": [V, V, S, I, I, I, V], "hobbit_commit_standard_host": [V, V, V, S, I, I, I, V], "hobbit_commitment_free": [V], "hobbit_commitment_num_leaves": [V],
        "hobbit_commitment_levels_dev": [V], "hobbit_commitment_tensor_dev": [V], "hobbit_commitment_levels": [V, V, V],
        "hobbit_commitment_root": [V, V, V], "hobbit_commitment_tensor_row": [V, V, I, I, V], "hobbit_commitment_gather": [V, V, V, V, S, V],
        "hobbit_commitment_path": [V, V, S, S, V], "hobbit_commitment_paths": [V, V, V, V, S, V], "hobbit_aggregate": [V, V, S, V, I, V],
        "hobbit_sumcheck2": [V, V, V, S, V, V, V, V, V], "hobbit_sumcheck3": [V, V, V, V, S, V, V, V, V, V],
        "hobbit_fill_splitmix": [V, V, S, U64],
        "hobbit_parity_matrix": [V, V, S, L, V], "hobbit_phi_g": [V, V, I, V, I, V], "hobbit_prepare_matrix_cols": [V, V, S, S, V, I, V],
        "hobbit_prove_linear_code": [V, V, S, L, V, V, V, V, V], "hobbit_prove_fft": [V, V, S, V, V, V, V, V],
        "hobbit_prove_fft_matrix": [V, V, S, S, V, V, V, V, V],
        "hobbit_gate_sumcheck": [V, V, V, V, V, V, V, S, V, V, V, V, V, V, V],
        "hobbit_open_core": [V, V, S, V, V, I, V], "hobbit_open_standard": [V, V, S, V, V, I, V], "hobbit_open_from_aggregate": [V, V, S, I, I, I, V], "hobbit_tensor_gather": [V, V, S, I, I, V, V, S, V], "hobbit_u64_bias_fold": [V, V, S, ctypes.c_uint64, I],
        "hobbit_whir_prove": [V, V, S, V, V, V, V], "hobbit_shockwave_prove": [V, V, V, V, S, I, V, I, V],
        "hobbit_shockwave_commit": [V, V, S, I, V, V], "hobbit_change_form": [V, V, I], "hobbit_whir_commit": [V, V, S, V, V],
        "hobbit_batch_3product_sumcheck": [V, V, V, V, V, I, V, V, V, V], "hobbit_mul_tree": [V, V, S, S, V, V, V, V, V, V, V, V, V, V],
        "hobbit_compute2p_error_terms": [V, V, V, V, V, S, V], "hobbit_compute3p_error_terms": [V, V, V, V, V, V, V, S, V],
        "hobbit_compute4p_error_terms": [V, V, V, V, V, V, V, V, V, S, V], "hobbit_fold_axpy": [V, V, V, V, S],
        "hobbit_fold_axpy_i32": [V, V, V, V, I, S], "hobbit_batch_prod": [V, V, V, V, V, V, V, I, S, V, V, V, V, V, V],
        "hobbit_elastic_begin": [V, S, I, I, I, V], "hobbit_elastic_push": [V, V, V], "hobbit_elastic_finish": [V, V, V],
        "hobbit_elastic_free": [V], "hobbit_elastic_push_inner": [V, V, V, V],
        "hobbit_elastic_open_begin": [V, S, S, I, V, I, V], "hobbit_elastic_open_aggregate_push": [V, V, V], "hobbit_elastic_open_aggregate_finish": [V, V],
        "hobbit_elastic_open_reply_push": [V, V, V], "hobbit_elastic_open_finish": [V, V, V, V], "hobbit_elastic_open_free": [V],
        "hobbit_elastic_open_begin_lin": [V, S, S, I, V, I, V], "hobbit_elastic_open_dims": [V, V, V, V],
        "hobbit_transcript_record": [I], "hobbit_transcript_read": [V, S],
        "hobbit_generate_randomness": [S, V],
        "hobbit_read_mul_tree_layer": [V, V, V, S, I, V], "hobbit_read_mul_tree_data": [V, V, V, S, I, I, I, V],
        "hobbit_generate_claims_opt": [V, V, V, S, S, V, I, I, I, I, V], "hobbit_sumcheck3_stream_batch": [V, V, V, S, S, V, I, I, I, I, V, I, V],
        "hobbit_mul_tree_stream_shallow": [V, V, V, S, S, I, S, V, I, V, I, V], "hobbit_gate_consistency_stream": [V, V, V, S, S, V, V],
        "hobbit_set_lookups": [V, I, V], "hobbit_open_standard_rs": [V, V, S, V, V, I, V], "hobbit_leaf_chain_relay": [V, V, S, I, I, I, S, S, V, V, V], "hobbit_leaf_chain": [V, V, S, I, I, I, V], "hobbit_verify_path_host": [V, ctypes.c_uint64, V, I, V, I], "hobbit_fingerprint_map": [V, V, V, V, V, V, V, S],
        "hobbit_axpy_aggregate": [V, V, V, V, S], "hobbit_stream_fold": [V, I, V, V, S, V], "hobbit_gate_consistency_lookups_stream": [V, V, V, S, S, V, V],
        "hobbit_tensorcode_chunks": [V, V, S, I, I, I, V], "hobbit_inner_digests": [V, V, S, I, I, V],
        "hobbit_chain_digests": [V, V, S, I, S, V], "hobbit_blake3_64_host": [V, V, S],
    }
    for name, args in protos.items():
        getattr(lib, name).argtypes = args
    lib.hobbit_transcript_count.restype = c_sz; lib.hobbit_transcript_read.restype = c_sz; lib.hobbit_transcript_record.restype = None
    return lib


def _hp(a):
    return a.ctypes.data_as(c_vp)


def Fh(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 2
    return a


class DeviceBuffer:
    """Device allocation owned through the C ABI (hobbit_malloc / hobbit_free)."""

    def __init__(self, hb, nbytes):
        self.hb = hb
        self.nbytes = int(nbytes)
        p = c_vp()
        hb._chk(hb.lib.hobbit_malloc(hb.ctx, c_sz(self.nbytes), ctypes.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr and self.hb.ctx:          # after Hobbit.close() the context (and its memory) is gone
            self.hb.lib.hobbit_free(self.hb.ctx, c_vp(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Commitment:
    """Handle on a device-resident Our_PC commitment (tensor + Merkle levels)."""

    def __init__(self, hb, handle, N, K, trs):
        self.hb, self.h, self.N, self.K, self.trs = hb, handle, N, K, trs
        self.M = N // K
        self.cols = 2 * self.M // trs

    def levels(self):
        out = np.zeros((2 * self.M - 1, 32), np.uint8)
        self.hb._chk(self.hb.lib.hobbit_commitment_levels(self.hb.ctx, self.h, _hp(out)))
        return out

    def root(self):
        out = np.zeros(32, np.uint8)
        self.hb._chk(self.hb.lib.hobbit_commitment_root(self.hb.ctx, self.h, _hp(out)))
        return out

    def tensor_row(self, chunk, row):
        out = np.zeros((self.cols, 2), np.uint64)
        self.hb._chk(self.hb.lib.hobbit_commitment_tensor_row(self.hb.ctx, self.h, c_int(chunk), c_int(row), _hp(out)))
        return out

    def tensor(self):
        """whole _tensor[K][2trs][cols] in the reference's layout (small cases only)"""
        return np.stack([np.stack([self.tensor_row(i, r) for r in range(2 * self.trs)]) for i in range(self.K)])

    def gather(self, rows, cols):
        """_compute_aggregation_reply: reply[q][i] = _tensor[i][rows[q]][cols[q]]"""
        rows = np.ascontiguousarray(rows, np.uint32); cols = np.ascontiguousarray(cols, np.uint32)
        out = np.zeros((len(rows), self.K, 2), np.uint64)
        self.hb._chk(self.hb.lib.hobbit_commitment_gather(self.hb.ctx, self.h, _hp(rows), _hp(cols), c_sz(len(rows)), _hp(out)))
        return out

    def open_tree_blake(self, col, row):
        depth = self.M.bit_length() - 1
        out = np.zeros((depth, 32), np.uint8)
        self.hb._chk(self.hb.lib.hobbit_commitment_path(self.hb.ctx, self.h, c_sz(col), c_sz(row), _hp(out)))
        return out

    def paths(self, cols, rows):
        cols = np.ascontiguousarray(cols, np.uint32); rows = np.ascontiguousarray(rows, np.uint32)
        depth = self.M.bit_length() - 1
        out = np.zeros((len(cols), depth, 32), np.uint8)
        self.hb._chk(self.hb.lib.hobbit_commitment_paths(self.hb.ctx, self.h, _hp(cols), _hp(rows), c_sz(len(cols)), _hp(out)))
        return out

    def free(self):
        if self.h and self.hb.ctx:
            self.hb.lib.hobbit_commitment_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Hobbit:
    """One context on one GPU.  Raises HobbitError when the library or the device is missing."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        ctx = c_vp()
        rc = self.lib.hobbit_ctx_create_on_stream(c_int(device), c_vp(stream) if stream else None, ctypes.byref(ctx))
        if rc != 0:
            raise HobbitError("hobbit_ctx_create failed (rc=%d): no usable HIP device -- there is no CPU fallback" % rc)
        self.ctx = ctx
        self._libc = ctypes.CDLL(None)
        self._libc.random.restype = ctypes.c_long

    def close(self):
        if self.ctx:
            self.lib.hobbit_ctx_destroy(self.ctx)
            self.ctx = None

    def _chk(self, rc):
        if rc != 0:
            raise HobbitError("rc=%d: %s" % (rc, self.lib.hobbit_last_error(self.ctx).decode()))

    # ---- memory
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        b = DeviceBuffer(self, max(arr.nbytes, 16))
        if arr.nbytes:
            self._chk(self.lib.hobbit_memcpy_h2d(self.ctx, c_vp(b.ptr), _hp(arr), c_sz(arr.nbytes)))
        return b

    def to_host(self, buf, shape, dtype, offset=0):
        out = np.zeros(shape, dtype)
        ptr = buf.ptr if isinstance(buf, DeviceBuffer) else int(buf)
        if out.nbytes:
            self._chk(self.lib.hobbit_memcpy_d2h(self.ctx, _hp(out), c_vp(ptr + offset), c_sz(out.nbytes)))
        return out

    def sync(self):
        self._chk(self.lib.hobbit_sync(self.ctx))

    # ---- timing / profiling
    def timer_begin(self):
        self._chk(self.lib.hobbit_timer_begin(self.ctx))

    def timer_end_ms(self):
        ms = ctypes.c_float()
        self._chk(self.lib.hobbit_timer_end_ms(self.ctx, ctypes.byref(ms)))
        return ms.value

    def profile(self, on=True):
        self._chk(self.lib.hobbit_profile_enable(self.ctx, c_int(int(on))))

    def profile_reset(self):
        self._chk(self.lib.hobbit_profile_reset(self.ctx))

    def profile_report(self):
        buf = ctypes.create_string_buffer(8192)
        self._chk(self.lib.hobbit_profile_names(self.ctx, buf, c_sz(8192)))
        out = {}
        for name in [s for s in buf.value.decode().split(";") if s]:
            ms = ctypes.c_double(); cnt = c_ll()
            self._chk(self.lib.hobbit_profile_get(self.ctx, name.encode(), ctypes.byref(ms), ctypes.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    # ---- field
    def f_binop(self, op, a, b):
        a, b = Fh(a), Fh(b)
        da, db = self.to_device(a), self.to_device(b)
        do = self.alloc(a.nbytes)
        self._chk(self.lib.hobbit_f_binop(self.ctx, c_int(op), c_vp(da.ptr), c_vp(db.ptr), c_vp(do.ptr), c_sz(a.size // 2)))
        return self.to_host(do, a.shape, np.uint64)

    def mimc_hash(self, x, k):
        x, k = Fh(x).reshape(2), Fh(k).reshape(2)
        o = np.zeros(2, np.uint64)
        self.lib.hobbit_mimc(_hp(x), _hp(k), _hp(o))
        return o

    def fill_splitmix(self, n, seed):
        b = self.alloc(16 * n)
        self._chk(self.lib.hobbit_fill_splitmix(self.ctx, c_vp(b.ptr), c_sz(n), ctypes.c_uint64(seed)))
        return b

    # ---- expander graphs: drawn on the host with libc rand()/random() in the reference's order
    def rng_reset(self):
        self._libc.srandom(1)

    def generate_randomness(self, n):
        """src/utils.cpp:873-883 (host-side, libc)"""
        out = np.zeros((n, 2), np.uint64)
        self.lib.hobbit_generate_randomness(c_sz(n), _hp(out))
        return out

    def _draw(self, L, R, d):
        nbr = np.zeros(L * d, np.int64); w = np.zeros((L * d, 2), np.uint64)
        rnd, rdm = self._libc.rand, self._libc.random
        for e in range(L * d):
            nbr[e] = rnd() % R          # src/expanders.h:36
            w[e, 0] = rdm()             # src/expanders.h:37
        return nbr, w

    def expander_init_store(self, n, weights=None):
        """src/expanders.h:78-92: draw _C[dep], recurse, draw D[dep]; upload every level.
        `weights`: optional dict {(dep,kind): F array} overriding the drawn weights (tests)."""
        self._chk(self.lib.hobbit_graph_reset(self.ctx))
        self._graph_levels = {}

        def rec(m, dep):
            if m <= 13:
                return m
            R = int(0.211 * m)
            self._graph_levels[(dep, 0)] = (m, R, 9) + self._draw(m, R, 9)
            L = rec(R, dep + 1)
            R2 = int(m * (1.72 - 1) - L)
            self._graph_levels[(dep, 1)] = (L, R2, 12) + self._draw(L, R2, 12)
            return m + L + R2

        total = rec(n, 0)
        for (dep, kind), (L, R, d, nbr, w) in self._graph_levels.items():
            if weights and (dep, kind) in weights:
                w = Fh(weights[(dep, kind)])
            self._chk(self.lib.hobbit_graph_upload(self.ctx, c_int(dep), c_int(kind), c_ll(L), c_ll(R), c_int(d), _hp(nbr), _hp(w)))
        ln = c_ll()
        self._chk(self.lib.hobbit_graph_finalize(self.ctx, c_ll(n), ctypes.byref(ln)))
        assert ln.value == total
        return total

    def upload_graphs(self, n, levels):
        """levels: {(dep,kind): dict(L,R,degree,nbr,w)} (e.g. taken from the oracle in tests)"""
        self._chk(self.lib.hobbit_graph_reset(self.ctx))
        for (dep, kind), g in levels.items():
            nbr = np.ascontiguousarray(g["nbr"], np.int64); w = Fh(g["w"])
            self._chk(self.lib.hobbit_graph_upload(self.ctx, c_int(dep), c_int(kind), c_ll(g["L"]), c_ll(g["R"]), c_int(g["degree"]), _hp(nbr), _hp(w)))
        ln = c_ll()
        self._chk(self.lib.hobbit_graph_finalize(self.ctx, c_ll(n), ctypes.byref(ln)))
        return ln.value

    def encode_monolithic(self, src, in_place=False):
        """src: (batch, n, 2) or (n, 2) -> (batch, 2n, 2).  in_place: the messages are laid out at the head of their 2n-element codewords and
        encoded there, as the commit does (the form the persistent deep-code kernels serve)."""
        s = Fh(src)
        single = s.ndim == 2
        if single:
            s = s[None]
        batch, n = s.shape[0], s.shape[1]
        if in_place:
            buf = np.zeros((batch, 2 * n, 2), np.uint64); buf[:, :n] = s
            buf[:, n:] = np.uint64(0x0123456789ABCDEF)                 # whatever the buffer held before must not matter
            dd = self.to_device(buf)
            self._chk(self.lib.hobbit_encode_batch(self.ctx, c_vp(dd.ptr), c_vp(dd.ptr), c_ll(n), c_sz(batch), c_sz(2 * n), c_sz(2 * n)))
            out = self.to_host(dd, (batch, 2 * n, 2), np.uint64)
            return out[0] if single else out
        ds = self.to_device(s); dd = self.alloc(batch * 2 * n * 16)
        self._chk(self.lib.hobbit_encode_batch(self.ctx, c_vp(ds.ptr), c_vp(dd.ptr), c_ll(n), c_sz(batch), c_sz(n), c_sz(2 * n)))
        out = self.to_host(dd, (batch, 2 * n, 2), np.uint64)
        return out[0] if single else out

    # ---- FFT
    def fft(self, arr, inverse=False):
        a = Fh(arr)
        single = a.ndim == 2
        if single:
            a = a[None]
        batch, ln = a.shape[0], a.shape[1]
        d = self.to_device(a)
        self._chk(self.lib.hobbit_fft_batch(self.ctx, c_vp(d.ptr), c_int(ln.bit_length() - 1), c_sz(batch), c_sz(ln), c_int(int(inverse))))
        out = self.to_host(d, a.shape, np.uint64)
        return out[0] if single else out

    # ---- hashes / Merkle
    def blake3_64(self, blocks):
        b = np.ascontiguousarray(blocks, np.uint8).reshape(-1, 64)
        d = self.to_device(b); o = self.alloc(32 * b.shape[0])
        self._chk(self.lib.hobbit_blake3_64(self.ctx, c_vp(d.ptr), c_vp(o.ptr), c_sz(b.shape[0])))
        return self.to_host(o, (b.shape[0], 32), np.uint8)

    def hash_md(self, xyzw, prev):
        x = Fh(xyzw).reshape(-1, 4, 2); p = np.ascontiguousarray(prev, np.uint8).reshape(-1, 32)
        dx, dp = self.to_device(x), self.to_device(p)
        self._chk(self.lib.hobbit_hash_md(self.ctx, c_vp(dx.ptr), c_vp(dp.ptr), c_vp(dp.ptr), c_sz(x.shape[0])))
        return self.to_host(dp, p.shape, np.uint8)

    def mt_commit_blake(self, leafs):
        x = Fh(leafs).reshape(-1, 2)
        n = x.shape[0] // 4
        d = self.to_device(x); lv = self.alloc(64 * n)
        self._chk(self.lib.hobbit_mt_commit_blake(self.ctx, c_vp(d.ptr), c_sz(x.shape[0]), c_vp(lv.ptr)))
        return self.to_host(lv, (2 * n - 1, 32), np.uint8)

    def create_tree_blake(self, level0, quirk=1):
        l0 = np.ascontiguousarray(level0, np.uint8).reshape(-1, 32)
        n = l0.shape[0]
        lv = self.alloc(64 * n)
        self._chk(self.lib.hobbit_memcpy_h2d(self.ctx, c_vp(lv.ptr), _hp(l0), c_sz(l0.nbytes)))
        self._chk(self.lib.hobbit_merkle_levels(self.ctx, c_vp(lv.ptr), c_sz(n), c_int(quirk)))
        return self.to_host(lv, (2 * n - 1, 32), np.uint8)

    # ---- multilinear
    def precompute_beta(self, r, keep_on_device=False):
        r = Fh(r).reshape(-1, 2)
        k = r.shape[0]
        d = self.alloc(16 << k)
        self._chk(self.lib.hobbit_eq_table(self.ctx, _hp(r), c_int(k), c_vp(d.ptr)))
        return d if keep_on_device else self.to_host(d, (1 << k, 2), np.uint64)

    def evaluate_vector(self, v, r):
        v = Fh(v).reshape(-1, 2); r = Fh(r).reshape(-1, 2)
        d = self.to_device(v); o = np.zeros(2, np.uint64)
        self._chk(self.lib.hobbit_eval_vector(self.ctx, c_vp(d.ptr), c_sz(v.shape[0]), _hp(r), _hp(o)))
        return o

    # ---- tensor code / commit
    def compute_tensorcode(self, msg, trs, lin):
        """returns the tensor in the reference's row-major (2trs, cols) layout"""
        m = Fh(msg).reshape(-1, 2)
        M = m.shape[0]; cols = 2 * M // trs
        d = self.to_device(m); o = self.alloc(16 * 4 * M)
        self._chk(self.lib.hobbit_tensorcode(self.ctx, c_vp(d.ptr), c_sz(M), c_int(trs), c_int(lin), c_vp(o.ptr)))
        t = self.to_host(o, (cols, 2 * trs, 2), np.uint64)      # codeword-major on the device
        return np.ascontiguousarray(t.transpose(1, 0, 2))

    def commit_standard(self, poly, K, trs, lin=1, sync=True):
        """poly: host array (N,2) or a DeviceBuffer/int pointer with N given as tuple (ptr, N).  sync=False returns once the commit is queued
        (the C call itself is asynchronous): whatever is queued next on this context -- an opening -- starts the moment the tree is done, without
        the host round trip in between; every accessor of the commitment synchronises as needed."""
        if isinstance(poly, tuple):
            ptr, N = poly
            ptr = ptr.ptr if isinstance(ptr, DeviceBuffer) else int(ptr)
            keep = None
        else:
            p = Fh(poly).reshape(-1, 2)
            N = p.shape[0]
            keep = self.to_device(p); ptr = keep.ptr
        h = c_vp()
        self._chk(self.lib.hobbit_commit_standard(self.ctx, c_vp(ptr), c_sz(N), c_int(K), c_int(trs), c_int(lin), ctypes.byref(h)))
        if sync or keep is not None:
            self.sync()
        return Commitment(self, h, N, K, trs)

    def commit_standard_host(self, poly, K, trs, lin=1):
        """commit_standard on a polynomial in (pageable) host memory, streamed to the device chunk group by chunk group under the commit's own
        kernels (hobbit_commit_standard_host).  Returns (Commitment, DeviceBuffer holding the uploaded polynomial)."""
        p = Fh(poly).reshape(-1, 2); N = p.shape[0]
        d = self.alloc(16 * N)
        h = c_vp()
        self._chk(self.lib.hobbit_commit_standard_host(self.ctx, _hp(p), c_vp(d.ptr), c_sz(N), c_int(K), c_int(trs), c_int(lin), ctypes.byref(h)))
        self.sync()
        return Commitment(self, h, N, K, trs), d

    def open_from_aggregate(self, aggr, K, trs, queries=5900):
        """open_standard's prover side from a given aggregate vector (host array or (DeviceBuffer/ptr, M)); multi-GPU open"""
        class _C:
            pass
        if isinstance(aggr, tuple):
            ptr, M = aggr; ptr = ptr.ptr if isinstance(ptr, DeviceBuffer) else int(ptr); keep = None
        else:
            a = Fh(aggr).reshape(-1, 2); M = a.shape[0]; keep = self.to_device(a); ptr = keep.ptr
        c = _C(); c.K = K; c.trs = trs; c.M = M; c.cols = 2 * M // trs; c.h = None
        return self.open_core((ptr, M), c, None, queries, want_paths=False, full=True, _from_aggregate=True)

    def open_core(self, poly, commitment, x, queries, want_paths=True, full=False, _from_aggregate=False):
        """open_standard + recursive_prover_Spielman (host: libc draws in the reference's order); full=False stops before the
        two shockwave_prove calls.  poly: host array or (DeviceBuffer, N)."""
        if isinstance(poly, tuple):
            ptr, N = poly; ptr = ptr.ptr if isinstance(ptr, DeviceBuffer) else int(ptr); keep = None
        else:
            p = Fh(poly).reshape(-1, 2); N = p.shape[0]; keep = self.to_device(p); ptr = keep.ptr
        c = commitment; x = Fh(x).reshape(-1, 2) if x is not None else None
        R1 = (2 * c.trs).bit_length() - 1; logc = c.cols.bit_length() - 1
        rounds = R1 + logc + 2 * (R1 + logc) + logc
        depth = c.M.bit_length() - 1
        Out = _OpenOut                    # (class creation costs ~0.1 ms: the ctypes mirrors of the two hot output structs are made once)
        res = dict(cols=np.zeros(queries, np.uint32), rows=np.zeros(queries, np.uint32), reply=None if _from_aggregate else np.zeros((queries, c.K, 2), np.uint64),
                   paths=np.zeros((queries, depth, 32), np.uint8) if want_paths else None, poly=np.zeros((rounds, 3, 2), np.uint64),
                   r=np.zeros((rounds, 2), np.uint64), vr=np.zeros((5, 2, 2), np.uint64), fin=np.zeros((5, 2), np.uint64),
                   scalars=np.zeros((5, 2), np.uint64), checks=np.zeros(3, np.int32), roots=np.zeros((2, 32), np.uint8))
        vals = [(res[k].ctypes.data if res[k] is not None else None) for k in ("cols", "rows", "reply", "paths", "poly", "r", "vr", "fin", "scalars", "checks", "roots")]
        sp = []
        if full:
            sp = [self._sp_buffers(c.trs * c.cols, 32), self._sp_buffers(c.M, 32)]
            vals += [ctypes.addressof(sp[0][1]), ctypes.addressof(sp[1][1])]
        else:
            vals += [None, None]
        o = Out(*vals)
        if _from_aggregate:
            self._chk(self.lib.hobbit_open_from_aggregate(self.ctx, ptr, N, c.K, c.trs, queries, ctypes.byref(o)))
        else:
            fn = self.lib.hobbit_open_standard if full else self.lib.hobbit_open_core
            self._chk(fn(self.ctx, ptr, N, c.h, _hp(x), queries, ctypes.byref(o)))
        res["I"] = np.stack([res["cols"], res["rows"]], axis=1)
        if full:
            res["sp_c"] = self._sp_trim(sp[0][0], c.trs * c.cols, 32); res["sp_f"] = self._sp_trim(sp[1][0], c.M, 32)
        return res

    def open_standard(self, poly, commitment, x, queries=5900, want_paths=True):
        """open_standard (src/Our_PC.cpp:604-661), prover side, including both shockwave_prove calls"""
        return self.open_core(poly, commitment, x, queries, want_paths, full=True)

    _SP_NAMES = ("I", "q1", "r1", "vr1", "fin1", "q2", "r2", "vr2", "fin2", "wq", "wa", "wroots", "wscal", "wchecks", "whir_root", "iters",
                 "reply", "paths", "qidx", "qreply", "qpaths", "final_pb", "qn")

    def _sp_buffers(self, N, k):
        w = N // k; W = 2 * w; lgW = W.bit_length() - 1; lw = w.bit_length() - 1
        out = dict(I=np.zeros(240, np.uint32), q1=np.zeros((lgW, 3, 2), np.uint64), r1=np.zeros((lgW, 2), np.uint64), vr1=np.zeros((2, 2), np.uint64),
                   fin1=np.zeros(2, np.uint64), q2=np.zeros((lgW, 3, 2), np.uint64), r2=np.zeros((lgW, 2), np.uint64), vr2=np.zeros((2, 2), np.uint64),
                   fin2=np.zeros(2, np.uint64), wq=np.zeros((lw + 8, 3, 2), np.uint64), wa=np.zeros((lw + 8, 2), np.uint64), wroots=np.zeros((lw + 1, 32), np.uint8),
                   wscal=np.zeros((2, 2), np.uint64), wchecks=np.zeros(2, np.int32), whir_root=np.zeros(32, np.uint8), iters=np.zeros(1, np.int32),
                   reply=np.zeros((240, k, 2), np.uint64), paths=np.zeros((240, lgW, 32), np.uint8))
        out.update(self._wq_buffers())
        return out, _ShockwaveOut(*[out[n].ctypes.data for n in self._SP_NAMES])

    @staticmethod
    def _sp_trim(out, N, k):
        lgW = (2 * N // k).bit_length() - 1; it = int(out["iters"][0])
        out["r2"] = out["r2"][:lgW - 1]; out["wq"] = out["wq"][:4 * it]; out["wa"] = out["wa"][:4 * it]; out["wroots"] = out["wroots"][:it]
        out["iters"] = np.array([it])
        if it:
            out.update(Hobbit._wq_trim(out, N // k, it))
        else:
            out.update(qn=np.zeros(0, np.int32), qidx=np.zeros(0, np.int32), qreply=np.zeros((0, 16, 2), np.uint64), qpaths=np.zeros(0, np.uint8), final_pb=np.zeros((0, 2), np.uint64))
        return out

    def aggregate(self, poly, beta):
        p = Fh(poly).reshape(-1, 2); b = Fh(beta).reshape(-1, 2)
        K = b.shape[0]
        d = self.to_device(p); o = self.alloc(16 * (p.shape[0] // K))
        self._chk(self.lib.hobbit_aggregate(self.ctx, c_vp(d.ptr), c_sz(p.shape[0]), _hp(b), c_int(K), c_vp(o.ptr)))
        return self.to_host(o, (p.shape[0] // K, 2), np.uint64)

    # ---- code-membership / FFT-as-sumcheck (reference names)
    def evaluate_parity_matrix(self, beta, n):
        b = Fh(beta).reshape(-1, 2)
        d = self.to_device(b); o = self.alloc(b.nbytes)
        self._chk(self.lib.hobbit_parity_matrix(self.ctx, d.ptr, b.shape[0], n, o.ptr))
        return self.to_host(o, b.shape, np.uint64)

    def phiGInit(self, rx, scale=(1, 0), ifft=False):
        r = Fh(rx).reshape(-1, 2); n = r.shape[0]
        sc = np.array(scale, np.uint64)
        o = self.alloc(16 << n)
        self._chk(self.lib.hobbit_phi_g(self.ctx, _hp(r), n, _hp(sc), int(ifft), o.ptr))
        return self.to_host(o, (1 << n, 2), np.uint64)

    def prepare_matrix_cols(self, M, r):
        m = Fh(M); rows, cols = m.shape[0], m.shape[1]; r = Fh(r).reshape(-1, 2)
        d = self.to_device(m); o = self.alloc(16 * cols)
        self._chk(self.lib.hobbit_prepare_matrix_cols(self.ctx, d.ptr, rows, cols, _hp(r), r.shape[0], o.ptr))
        return self.to_host(o, (cols, 2), np.uint64)

    def _proof2(self, rounds):
        return (np.zeros((rounds, 3, 2), np.uint64), np.zeros((rounds, 2), np.uint64), np.zeros((2, 2), np.uint64), np.zeros(2, np.uint64))

    def prove_linear_code(self, codeword, n, r1):
        cw = Fh(codeword).reshape(-1, 2); r1 = Fh(r1).reshape(-1, 2)
        q, r, vr, fin = self._proof2(cw.shape[0].bit_length() - 1)
        d = self.to_device(cw)
        self._chk(self.lib.hobbit_prove_linear_code(self.ctx, d.ptr, cw.shape[0], n, _hp(r1), _hp(q), _hp(r), _hp(vr), _hp(fin)))
        return dict(poly=q, r=r, vr=vr, fin=fin)

    def prove_fft(self, m, rr):
        m = Fh(m).reshape(-1, 2); rr = Fh(rr).reshape(-1, 2)
        rounds = (2 * m.shape[0]).bit_length() - 1
        q, r, vr, fin = self._proof2(rounds)
        d = self.to_device(m)
        self._chk(self.lib.hobbit_prove_fft(self.ctx, d.ptr, m.shape[0], _hp(rr), _hp(q), _hp(r), _hp(vr), _hp(fin)))
        return dict(poly=q, r=r[:rounds - 1], vr=vr, fin=fin)       # the reference pops the last challenge (src/sumcheck.cpp:2985)

    def prove_fft_matrix(self, M, rr):
        m = Fh(M); rows, cols = m.shape[0], m.shape[1]; rr = Fh(rr).reshape(-1, 2)
        q, r, vr, fin = self._proof2((2 * cols).bit_length() - 1)
        d = self.to_device(m)
        self._chk(self.lib.hobbit_prove_fft_matrix(self.ctx, d.ptr, rows, cols, _hp(rr), _hp(q), _hp(r), _hp(vr), _hp(fin)))
        return dict(poly=q, r=r, vr=vr, fin=fin)

    # ---- inner PCS commitments of the opening (src/Virgo.cpp:104-178)
    def shockwave_commit(self, poly, k):
        p = Fh(poly).reshape(-1, 2); N = p.shape[0]; W = 2 * N // k
        d = self.to_device(p); enc = self.alloc(16 * k * W); lv = self.alloc(64 * W)
        self._chk(self.lib.hobbit_shockwave_commit(self.ctx, d.ptr, N, k, enc.ptr, lv.ptr))
        return self.to_host(enc, (k, W, 2), np.uint64), self.to_host(lv, (2 * W - 1, 32), np.uint8)

    def change_form(self, poly):
        p = Fh(poly).reshape(-1, 2)
        d = self.to_device(p)
        self._chk(self.lib.hobbit_change_form(self.ctx, d.ptr, p.shape[0].bit_length() - 1))
        return self.to_host(d, p.shape, np.uint64)

    def whir_commit(self, poly):
        p = Fh(poly).reshape(-1, 2); N = p.shape[0]
        d = self.to_device(p); com = self.alloc(32 * N); lv = self.alloc(32 * N)
        self._chk(self.lib.hobbit_whir_commit(self.ctx, d.ptr, N, com.ptr, lv.ptr))
        return self.to_host(com, (2 * N, 2), np.uint64), self.to_host(lv, (N - 1, 32), np.uint8)

    _WQ_NAMES = ("qidx", "qreply", "qpaths", "final_pb", "qn")

    @staticmethod
    def _wq_buffers():
        return dict(qidx=np.zeros(256, np.int32), qreply=np.zeros((256, 16, 2), np.uint64), qpaths=np.zeros(256 * 24 * 32, np.uint8),
                    final_pb=np.zeros((32, 2), np.uint64), qn=np.zeros(8, np.int32))

    @staticmethod
    def _wq_trim(b, N, iters):
        nq = b["qn"][:iters].copy(); tot = int(nq.sum())
        pbytes = sum(int(nq[t]) * 32 * (((2 * N) >> t).bit_length() - 1 - 2) for t in range(iters))
        rem = N >> (4 * iters)
        return dict(qn=nq, qidx=b["qidx"][:tot].copy(), qreply=b["qreply"][:tot].copy(), qpaths=b["qpaths"][:pbytes].copy(), final_pb=b["final_pb"][:2 * rem].copy())

    def whir_prove(self, poly, x, com=None, com_levels=None):
        """_whir_prove (src/Virgo.cpp:519-686), prover side; com / com_levels: device buffers from whir_commit (computed here if None)"""
        p = Fh(poly).reshape(-1, 2); x = Fh(x).reshape(-1, 2); N = p.shape[0]; logN = N.bit_length() - 1
        out = dict(qpoly=np.zeros((logN + 8, 3, 2), np.uint64), a=np.zeros((logN + 8, 2), np.uint64), fri_roots=np.zeros((logN, 32), np.uint8),
                   scal=np.zeros((2, 2), np.uint64), checks=np.zeros(2, np.int32), iters=np.zeros(1, np.int32))
        out.update(self._wq_buffers())
        names = ("qpoly", "a", "fri_roots", "scal", "checks", "iters") + self._WQ_NAMES

        class Out(ctypes.Structure):
            _fields_ = [(n, c_vp) for n in names]
        o = Out(*[out[n].ctypes.data for n in names])
        d = self.to_device(p)
        dc, dl = self.alloc(16 * 2 * N), self.alloc(32 * N)
        self._chk(self.lib.hobbit_whir_commit(self.ctx, d.ptr, N, dc.ptr, dl.ptr))
        self._chk(self.lib.hobbit_whir_prove(self.ctx, d.ptr, N, dc.ptr, dl.ptr, _hp(x), ctypes.byref(o)))
        it = int(out["iters"][0])
        res = dict(iters=np.array([it]), poly=out["qpoly"][:4 * it], a=out["a"][:4 * it], roots=out["fri_roots"][:it], scal=out["scal"], checks=out["checks"])
        res.update(self._wq_trim(out, N, it))
        return res

    def shockwave_prove(self, matrix, enc, k, x, levels=None):
        m = Fh(matrix).reshape(-1, 2); e = Fh(enc).reshape(-1, 2); x = Fh(x).reshape(-1, 2)
        N = m.shape[0]
        out, o = self._sp_buffers(N, k)
        dm, de = self.to_device(m), self.to_device(e)
        dl = self.to_device(np.ascontiguousarray(levels, np.uint8)) if levels is not None else None
        self._chk(self.lib.hobbit_shockwave_prove(self.ctx, dm.ptr, de.ptr, dl.ptr if dl is not None else None, N, k, _hp(x), x.shape[0], ctypes.byref(o)))
        return self._sp_trim(out, N, k)

    # ---- batched cubic sumcheck / multiplication tree (src/sumcheck.cpp:275-372, 35-257)
    def batch_3product_sumcheck(self, t1, t2, t3, lens, a):
        T = [Fh(x).reshape(-1, 2) for x in (t1, t2, t3)]
        d = [self.to_device(x) for x in T]
        lens = np.ascontiguousarray(lens, np.uint64); av = Fh(a).reshape(-1, 2)
        rounds = int(max(lens)).bit_length() - 1; nb = len(lens)
        q = np.zeros((rounds, 4, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64); vr = np.zeros((nb, 3, 2), np.uint64)
        self._chk(self.lib.hobbit_batch_3product_sumcheck(self.ctx, d[0].ptr, d[1].ptr, d[2].ptr, _hp(lens), nb, _hp(av), _hp(q), _hp(r), _hp(vr)))
        return dict(poly=q, r=r, vr=vr)

    def mul_tree(self, inp, previous_r, prev_x=None):
        x = Fh(inp); vectors, size = x.shape[0], x.shape[1]
        lt = (vectors * size).bit_length() - 1; depth = size.bit_length() - 1
        nr = sum(range(lt))
        q = np.zeros((nr + 1, 4, 2), np.uint64); r = np.zeros((nr + 1, 2), np.uint64)
        vr = np.zeros((depth, 3, 2), np.uint64); fin = np.zeros((depth, 2), np.uint64)
        final_r = np.zeros((lt, 2), np.uint64); oe = np.zeros(2, np.uint64); fe = np.zeros(2, np.uint64)
        pr = Fh(previous_r).reshape(2); px = Fh(prev_x).reshape(-1, 2) if prev_x is not None else None
        d = self.to_device(x); layers = ctypes.c_int()
        self._chk(self.lib.hobbit_mul_tree(self.ctx, d.ptr, vectors, size, _hp(pr), _hp(px) if px is not None else None, _hp(q), _hp(r), _hp(vr), _hp(fin),
                                           _hp(final_r), _hp(oe), _hp(fe), ctypes.byref(layers)))
        L = layers.value
        return dict(layers=np.array([L]), poly=q, r=r, vr=vr[:L], fin=fin[:L], final_r=final_r, out_eval=oe, final_eval=fe)

    # ---- streaming-sumcheck error terms / folds (reference names: src/sumcheck.cpp:374-432, 1093-1136)
    def err2p(self, b1, b2, f1, f2):
        d = [self.to_device(Fh(x).reshape(-1, 2)) for x in (b1, b2, f1, f2)]
        K = np.zeros((2, 2), np.uint64)
        self._chk(self.lib.hobbit_compute2p_error_terms(self.ctx, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, len(Fh(b1).reshape(-1, 2)), _hp(K)))
        return K

    def err3p(self, b1, gate, f1, f2, f3, beta):
        g = self.to_device(np.ascontiguousarray(gate, np.int32))
        d = [self.to_device(Fh(x).reshape(-1, 2)) for x in (b1, f1, f2, f3, beta)]
        K = np.zeros((3, 2), np.uint64)
        self._chk(self.lib.hobbit_compute3p_error_terms(self.ctx, d[0].ptr, g.ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, len(gate), _hp(K)))
        return K

    def err4p(self, b1, b2, b3, gate, f1, f2, f3, f4):
        g = self.to_device(np.ascontiguousarray(gate, np.int32))
        d = [self.to_device(Fh(x).reshape(-1, 2)) for x in (b1, b2, b3, f1, f2, f3, f4)]
        K = np.zeros((4, 2), np.uint64)
        self._chk(self.lib.hobbit_compute4p_error_terms(self.ctx, d[0].ptr, d[1].ptr, d[2].ptr, g.ptr, d[3].ptr, d[4].ptr, d[5].ptr, d[6].ptr, len(gate), _hp(K)))
        return K

    def batch_prod(self, f1, f2, f3, b1, b2, b3, r_last, a, rem_beta, Kf, Kp):
        f = [Fh(x) for x in (f1, f2, f3)]; b = [Fh(x) for x in (b1, b2, b3)]
        batches, n = f[0].shape[0], f[0].shape[1]
        df = [self.to_device(x) for x in f]; db = [self.to_device(x) for x in b]
        rl = Fh(r_last).reshape(2); av = Fh(a).reshape(-1, 2); rb = Fh(rem_beta).reshape(-1, 2)
        kf = Fh(Kf).reshape(2).copy(); kp = Fh(Kp).reshape(-1, 2).copy(); ro = np.zeros(2, np.uint64)
        self._chk(self.lib.hobbit_batch_prod(self.ctx, df[0].ptr, df[1].ptr, df[2].ptr, db[0].ptr, db[1].ptr, db[2].ptr, batches, n, _hp(rl), _hp(av), _hp(rb),
                                             _hp(kf), _hp(kp), _hp(ro)))
        return dict(rand=ro, Kf=kf, Kp=kp, f1=self.to_host(df[0], f[0].shape, np.uint64), f2=self.to_host(df[1], f[0].shape, np.uint64),
                    f3=self.to_host(df[2], f[0].shape, np.uint64))

    # ---- Elastic_PC streaming commit (src/Elastic_PC.cpp:174-285) on the synthetic "test" stream
    def read_stream_PC(self, B):
        """default branch of read_stream_PC (src/witness_stream.cpp:2405-2411): n = 322322; v[i] = n; n = n*n + i.
        Host-side and sequential, as in the reference; every chunk of the "test" stream is identical."""
        out = np.zeros((B, 2), np.uint64)
        n = np.array([[322322, 0]], np.uint64)
        one = np.zeros((1, 2), np.uint64)
        lib = self.lib
        tmp = np.zeros((1, 2), np.uint64)
        for i in range(B):
            out[i] = n[0]
            lib.hobbit_f_mul_host(_hp(n), _hp(n), _hp(tmp), 1)
            one[0, 0] = i
            r = (int(tmp[0, 0]) + i) % P
            n[0, 0] = r; n[0, 1] = tmp[0, 1]
        return out

    def read_stream(self, B):
        """default branch of read_stream (src/witness_stream.cpp:2348-2352): v[i] = F(i % 1024 + 1); what Elastic_PC::open's two
        passes read for the "test" descriptor"""
        out = np.zeros((B, 2), np.uint64)
        out[:, 0] = np.arange(B, dtype=np.uint64) % np.uint64(1024) + np.uint64(1)
        return out

    def elastic_open(self, N, B, x, queries=700, commit_levels=None, chunk=None, shockwave=True):
        """Elastic_PC::open, option 1 (src/Elastic_PC.cpp:625-726), prover side.  commit_levels: DeviceBuffer with the commitment tree
        (hobbit_elastic_finish) or None; chunk: DeviceBuffer of the (repeating) read_stream chunk, built here if None."""
        x = Fh(x).reshape(-1, 2)
        trs = B >> 11; cols = 4096; K = N // B
        logc = 12; logr = (2 * trs).bit_length() - 1; logt = logr - 1
        e = c_vp()
        self._chk(self.lib.hobbit_elastic_open_begin(self.ctx, N, B, trs, _hp(x), queries, ctypes.byref(e)))
        try:
            if chunk is None:
                chunk = self.to_device(self.read_stream(B))
            for _ in range(K):
                self._chk(self.lib.hobbit_elastic_open_aggregate_push(self.ctx, e, chunk.ptr))
            self._chk(self.lib.hobbit_elastic_open_aggregate_finish(self.ctx, e))
            for _ in range(K):
                self._chk(self.lib.hobbit_elastic_open_reply_push(self.ctx, e, chunk.ptr))
            maxr = (2048 * 2 * trs).bit_length() - 1 + logr + (logt + logc) + logc
            depth = (4 * B).bit_length() - 1
            res = dict(cols=np.zeros(queries, np.uint32), rows=np.zeros(queries, np.uint32), rv0=np.zeros(2, np.uint64), reply=np.zeros((queries, K, 2), np.uint64),
                       reply_len=np.zeros(1, np.int32), paths=np.zeros((queries, depth, 32), np.uint8) if commit_levels is not None else None,
                       cf_root=np.zeros(32, np.uint8), ncols=np.zeros(1, np.int32), poly=np.zeros((maxr, 3, 2), np.uint64), r=np.zeros((maxr, 2), np.uint64),
                       vr=np.zeros((4, 2, 2), np.uint64), fin=np.zeros((4, 2), np.uint64), checks=np.zeros(2, np.int32), rx=np.zeros((logc + logt, 2), np.uint64))
            names = ("cols", "rows", "rv0", "reply", "reply_len", "paths", "cf_root", "ncols", "poly", "r", "vr", "fin", "checks", "rx")

            class Out(ctypes.Structure):
                _fields_ = [(n, c_vp) for n in names + ("sp_f",)]
            sp = self._sp_buffers(B, 32) if shockwave else None
            o = Out(*([(res[k].ctypes.data if res[k] is not None else None) for k in names] + [ctypes.addressof(sp[1]) if sp else None]))
            lv = commit_levels.ptr if isinstance(commit_levels, DeviceBuffer) else commit_levels
            self._chk(self.lib.hobbit_elastic_open_finish(self.ctx, e, lv, ctypes.byref(o)))
        finally:
            self.lib.hobbit_elastic_open_free(e)
        nc = int(res["ncols"][0]); np2 = 1 << max(nc - 1, 0).bit_length()
        rounds = (np2 * 2 * trs).bit_length() - 1 + logr + (logt + logc) + logc
        res["poly"] = res["poly"][:rounds]; res["r"] = res["r"][:rounds]
        rl = int(res["reply_len"][0])
        res["reply"] = res["reply"].reshape(-1, 2)[:queries * rl].reshape(queries, rl, 2)
        res["I"] = np.stack([res["cols"], res["rows"]], axis=1)
        if sp:
            res["sp_f"] = self._sp_trim(sp[0], B, 32)
        return res

    def elastic_open2(self, N, B, x, queries=5900, commit_levels=None, chunks=None):
        """Elastic_PC::open, option 2 (RS x expander; src/Elastic_PC.cpp:625-726 under linear_time), prover side.  The expander graphs for
        tensor_row_size = B >> 14 must have been uploaded (upload_graphs).  chunks: None = the reference's repeating read_stream chunk; or a
        callable i -> host array (B, 2) / DeviceBuffer, called once per pass and chunk in stream order."""
        x = Fh(x).reshape(-1, 2)
        trs = B >> 14; cols = 2 * B // trs; K = N // B
        logc = cols.bit_length() - 1; R1 = (2 * trs).bit_length() - 1; logt = R1 - 1
        e = c_vp()
        self._chk(self.lib.hobbit_elastic_open_begin_lin(self.ctx, N, B, trs, _hp(x), queries, ctypes.byref(e)))
        try:
            nc, nr, npad = c_int(), c_int(), c_sz()
            self.lib.hobbit_elastic_open_dims(e, ctypes.byref(nc), ctypes.byref(nr), ctypes.byref(npad))
            nc, nr, npad = nc.value, nr.value, npad.value
            resident = self.to_device(self.read_stream(B)) if chunks is None else None

            def chunk(i):
                if resident is not None:
                    return resident
                c = chunks(i)
                return c if isinstance(c, DeviceBuffer) else self.to_device(Fh(c).reshape(-1, 2))
            for i in range(K):
                self._chk(self.lib.hobbit_elastic_open_aggregate_push(self.ctx, e, chunk(i).ptr))
            self._chk(self.lib.hobbit_elastic_open_aggregate_finish(self.ctx, e))
            for i in range(K):
                self._chk(self.lib.hobbit_elastic_open_reply_push(self.ctx, e, chunk(i).ptr))
            R3 = npad.bit_length() - 1
            rounds = R1 + logc + R3 + logc
            depth = (4 * B).bit_length() - 1
            res = dict(cols=np.zeros(queries, np.uint32), rows=np.zeros(queries, np.uint32), rv0=np.zeros(2, np.uint64), reply=np.zeros((queries, K, 2), np.uint64),
                       reply_len=np.zeros(1, np.int32), paths=np.zeros((queries, depth, 32), np.uint8) if commit_levels is not None else None,
                       cf_root=np.zeros(32, np.uint8), ncols=np.zeros(1, np.int32), poly=np.zeros((rounds, 3, 2), np.uint64), r=np.zeros((rounds, 2), np.uint64),
                       vr=np.zeros((4, 2, 2), np.uint64), fin=np.zeros((4, 2), np.uint64), checks=np.zeros(1, np.int32), rx=np.zeros((logc + logt - 1, 2), np.uint64),
                       cc_root=np.zeros(32, np.uint8), nr=np.zeros(1, np.int32), scal=np.zeros((3, 2), np.uint64))
            names = ("cols", "rows", "rv0", "reply", "reply_len", "paths", "cf_root", "ncols", "poly", "r", "vr", "fin", "checks", "rx")
            names2 = ("cc_root", "nr", "scal")

            class Out(ctypes.Structure):
                _fields_ = [(n, c_vp) for n in names + ("sp_f",) + names2 + ("sp_c",)]
            spf = self._sp_buffers(B, 32); spc = self._sp_buffers(npad, 32)
            o = Out(*([(res[k].ctypes.data if res[k] is not None else None) for k in names] + [ctypes.addressof(spf[1])] +
                      [res[k].ctypes.data for k in names2] + [ctypes.addressof(spc[1])]))
            lv = commit_levels.ptr if isinstance(commit_levels, DeviceBuffer) else commit_levels
            self._chk(self.lib.hobbit_elastic_open_finish(self.ctx, e, lv, ctypes.byref(o)))
        finally:
            self.lib.hobbit_elastic_open_free(e)
        rl = int(res["reply_len"][0])
        res["reply"] = res["reply"].reshape(-1, 2)[:queries * rl].reshape(queries, rl, 2)
        res["I"] = np.stack([res["cols"], res["rows"]], axis=1)
        res["sp_f"] = self._sp_trim(spf[0], B, 32); res["sp_c"] = self._sp_trim(spc[0], npad, 32)
        return res

    def open_standard_rs(self, poly, c, x, queries=790, want_paths=True, shockwave=True):
        """Our_PC open_standard with linear_time == false (test_PC option 1, src/Our_PC.cpp:604-692), prover side.  poly: host array or
        (DeviceBuffer/ptr, N); c: the Commitment from commit_standard(..., lin=0)."""
        if isinstance(poly, tuple):
            ptr, N = poly; ptr = ptr.ptr if isinstance(ptr, DeviceBuffer) else int(ptr); keep = None
        else:
            p = Fh(poly).reshape(-1, 2); N = p.shape[0]; keep = self.to_device(p); ptr = keep.ptr
        x = Fh(x).reshape(-1, 2)
        K, trs, B = c.K, c.trs, c.M
        cols = 2 * B // trs; logc = cols.bit_length() - 1; logr = (2 * trs).bit_length() - 1; logt = logr - 1
        maxr = (2048 * 2 * trs).bit_length() - 1 + logr + (logt + logc) + logc
        depth = B.bit_length() - 1
        res = dict(cols=np.zeros(queries, np.uint32), rows=np.zeros(queries, np.uint32), rv0=np.zeros(2, np.uint64), reply=np.zeros((queries, K, 2), np.uint64),
                   reply_len=np.zeros(1, np.int32), paths=np.zeros((queries, depth, 32), np.uint8) if want_paths else None,
                   cf_root=np.zeros(32, np.uint8), ncols=np.zeros(1, np.int32), poly=np.zeros((maxr, 3, 2), np.uint64), r=np.zeros((maxr, 2), np.uint64),
                   vr=np.zeros((4, 2, 2), np.uint64), fin=np.zeros((4, 2), np.uint64), checks=np.zeros(2, np.int32), rx=np.zeros((logc + logt, 2), np.uint64))
        names = ("cols", "rows", "rv0", "reply", "reply_len", "paths", "cf_root", "ncols", "poly", "r", "vr", "fin", "checks", "rx")

        class Out(ctypes.Structure):
            _fields_ = [(n, c_vp) for n in names + ("sp_f",)]
        sp = self._sp_buffers(B, 32) if shockwave else None
        o = Out(*([(res[k].ctypes.data if res[k] is not None else None) for k in names] + [ctypes.addressof(sp[1]) if sp else None]))
        self._chk(self.lib.hobbit_open_standard_rs(self.ctx, c_vp(ptr), c_sz(N), c.h, _hp(x), c_int(queries), ctypes.byref(o)))
        nc = int(res["ncols"][0]); np2 = 1 << max(nc - 1, 0).bit_length()
        rounds = (np2 * 2 * trs).bit_length() - 1 + logr + (logt + logc) + logc
        res["poly"] = res["poly"][:rounds]; res["r"] = res["r"][:rounds]
        res["I"] = np.stack([res["cols"], res["rows"]], axis=1)
        if sp:
            res["sp_f"] = self._sp_trim(sp[0], B, 32)
        return res

    # ---- streaming provers over a chunk source (src/sumcheck.cpp:796-975, 1014-1054, 1150-1393, 1746-1915) ----
    CHUNK_FN = ctypes.CFUNCTYPE(c_int, c_vp, c_sz, ctypes.POINTER(c_vp))
    TRACE_FN = ctypes.CFUNCTYPE(c_int, c_vp, c_sz, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp))

    def chunk_source(self, kind=0, seed=0):
        """A hobbit_chunk_source for tests / benches.  kind 0: the reference's default stream (read_stream: v[i] = F(i % 1024 + 1), every
        read alike; one resident buffer per read size).  kind 1: read c since the last reset = splitmix_field(n, seed + c), generated on
        the host and uploaded per read.  Keep the returned object alive while the library may call it."""
        st = dict(c=0, resident={}, live=[])

        def fn(user, n, out):
            try:
                if n == 0:
                    st["c"] = 0
                    return 0
                if kind == 0:
                    if n not in st["resident"]:
                        st["resident"][n] = self.to_device(self.read_stream(n))
                    out[0] = st["resident"][n].ptr
                else:
                    buf = self.to_device(splitmix_field(n, seed + st["c"])); st["c"] += 1
                    st["live"].append(buf); del st["live"][:-3]
                    out[0] = buf.ptr
                return 0
            except Exception:            # an exception must not unwind through the C caller
                import traceback; traceback.print_exc()
                return 1
        cb = self.CHUNK_FN(fn); cb._state = st
        return cb

    def trace_source(self, L, R, O, S, B):
        """hobbit_trace_source over host arrays (chunk c = elements [c*B, (c+1)*B)), resident on the device"""
        dl, dr, do = [self.to_device(Fh(v).reshape(-1, 2)) for v in (L, R, O)]; ds = self.to_device(np.ascontiguousarray(S, np.int32))
        st = dict(c=0)

        def fn(user, n, pl, pr, po, ps):
            if n == 0:
                st["c"] = 0
                return 0
            c = st["c"]; st["c"] += 1
            if n != B or (c + 1) * B * 16 > dl.nbytes:
                return 1
            pl[0] = dl.ptr + 16 * c * B; pr[0] = dr.ptr + 16 * c * B; po[0] = do.ptr + 16 * c * B; ps[0] = ds.ptr + 4 * c * B
            return 0
        cb = self.TRACE_FN(fn); cb._keep = (dl, dr, do, ds)
        return cb

    def read_mul_tree_layer(self, src, size, layer):
        o = self.alloc(16 * size)
        self._chk(self.lib.hobbit_read_mul_tree_layer(self.ctx, ctypes.cast(src, c_vp), None, size, layer, o.ptr))
        return self.to_host(o, (size, 2), np.uint64)

    def read_mul_tree_data(self, src, size, layer, distance, batches):
        tot = sum(size >> (i * distance) for i in range(batches))
        o = self.alloc(16 * tot)
        self._chk(self.lib.hobbit_read_mul_tree_data(self.ctx, ctypes.cast(src, c_vp), None, size, layer, distance, batches, o.ptr))
        return self.to_host(o, (tot, 2), np.uint64)

    def generate_claims_opt(self, src, fd_size, B, r, batches, layer_id, distance):
        r = Fh(r).reshape(-1, 2); c = np.zeros((batches, 2), np.uint64)
        self._chk(self.lib.hobbit_generate_claims_opt(self.ctx, ctypes.cast(src, c_vp), None, fd_size, B, _hp(r), r.shape[0], batches, layer_id, distance, _hp(c)))
        return c

    class _S3(ctypes.Structure):
        _fields_ = [("new_claims", c_vp), ("new_r", c_vp), ("new_r_ld", c_int), ("cpoly1", c_vp), ("r1", c_vp), ("vr1", c_vp), ("qpoly2", c_vp), ("r2", c_vp), ("vr2", c_vp),
                    ("fin2", c_vp), ("R", c_vp), ("checks", c_vp)]

    def _s3_buffers(self, fd_size, B, batches, layer_id):
        size = fd_size >> layer_id; logB = B.bit_length() - 1; nR = size // (2 * B); lR = nR.bit_length() - 1; ld = 1 + logB + lR
        b = dict(new_claims=np.zeros((batches, 2), np.uint64), new_r=np.zeros((batches, ld, 2), np.uint64), cpoly1=np.zeros((logB, 4, 2), np.uint64),
                 r1=np.zeros((logB, 2), np.uint64), vr1=np.zeros((batches, 3, 2), np.uint64), qpoly2=np.zeros((lR, 3, 2), np.uint64), r2=np.zeros((lR, 2), np.uint64),
                 vr2=np.zeros((2, 2), np.uint64), fin2=np.zeros(2, np.uint64), R=np.zeros((nR, 2), np.uint64), checks=np.zeros(3, np.int32))
        st = self._S3(b["new_claims"].ctypes.data, b["new_r"].ctypes.data, ld, *[b[k].ctypes.data for k in ("cpoly1", "r1", "vr1", "qpoly2", "r2", "vr2", "fin2", "R", "checks")])
        return b, st, ld

    @staticmethod
    def _s3_result(b, B, batches, distance, ld):
        logB = B.bit_length() - 1; lR = ld - 1 - logB
        out = dict(b); out["new_r"] = [b["new_r"][i, :1 + logB - i * distance + lR].copy() for i in range(batches)]
        return out

    def sumcheck3_stream_batch(self, src, fd_size, B, r, batches, distance, layer_id, old_claims):
        """generate_3product_sumcheck_beta_stream_batch_optimized; r: (batches, rlen, 2)"""
        r = Fh(r); rlen = r.shape[1]; oc = Fh(old_claims).reshape(-1, 2)
        b, st, ld = self._s3_buffers(fd_size, B, batches, layer_id)
        self._chk(self.lib.hobbit_sumcheck3_stream_batch(self.ctx, ctypes.cast(src, c_vp), None, fd_size, B, _hp(r), rlen, batches, distance, layer_id, _hp(oc), oc.shape[0], ctypes.byref(st)))
        return self._s3_result(b, B, batches, distance, ld)

    def mul_tree_stream_shallow(self, src, fd_size, B, vectors, size, previous_r, distance, prev_x, naive=True):
        """prove_multiplication_tree_stream_shallow without commit_layers / open_layers"""
        total = size * vectors
        layers = max((total // (2 * B)).bit_length() - 1, 0)
        if layers % distance != 0 and layers > distance:
            layers = distance + layers - (layers % distance)
        n1 = (fd_size >> layers) if total > 2 * B else total
        sz = n1 // vectors; lt = n1.bit_length() - 1; depth = sz.bit_length() - 1; nr = sum(range(lt))
        tree = dict(output=np.zeros((vectors, 2), np.uint64), poly=np.zeros((nr + 1, 4, 2), np.uint64), r=np.zeros((nr + 1, 2), np.uint64), vr=np.zeros((depth, 3, 2), np.uint64),
                    fin=np.zeros((depth, 2), np.uint64), final_r=np.zeros((lt, 2), np.uint64), out_eval=np.zeros(2, np.uint64), final_eval=np.zeros(2, np.uint64),
                    layers=np.zeros(1, np.int32))
        if layers <= distance or naive:
            plan = [(1, 1, i) for i in range(layers - 1, -1, -1)]
        else:
            plan = [(layers // distance, distance, i) for i in range(distance - 1, -1, -1)]
        bufs = [self._s3_buffers(fd_size, B, bt, lid) for (bt, d, lid) in plan] if total > 2 * B else []
        arr = (self._S3 * max(len(bufs), 1))(*[x[1] for x in bufs])
        nst = ctypes.c_int(0); claims0 = np.zeros((16, 2), np.uint64); sl = ctypes.c_int(0)

        class MO(ctypes.Structure):
            _fields_ = [(n, c_vp) for n in ("output", "cpoly", "r", "vr", "fin", "final_r", "out_eval", "final_eval", "layers", "steps")] + [("max_steps", c_int), ("n_steps", c_vp), ("claims0", c_vp), ("stream_layers", c_vp)]
        mo = MO(*[tree[k].ctypes.data for k in ("output", "poly", "r", "vr", "fin", "final_r", "out_eval", "final_eval", "layers")], ctypes.addressof(arr), len(bufs),
                ctypes.addressof(nst), claims0.ctypes.data, ctypes.addressof(sl))
        pr = Fh(previous_r).reshape(2); px = Fh(prev_x).reshape(-1, 2)
        self._chk(self.lib.hobbit_mul_tree_stream_shallow(self.ctx, ctypes.cast(src, c_vp), None, fd_size, B, vectors, size, _hp(pr), distance, _hp(px), int(naive), ctypes.byref(mo)))
        L = int(tree["layers"][0])
        tree.update(vr=tree["vr"][:L], fin=tree["fin"][:L], layers=np.array([L]))
        steps = [self._s3_result(bufs[i][0], B, plan[i][0], plan[i][1], bufs[i][2]) for i in range(nst.value)]
        return dict(output=tree["output"], tree=tree, steps=steps, layers=sl.value, claims0=claims0)

    def gate_consistency_stream(self, tsrc, n_chunks, B, r):
        r = Fh(r).reshape(-1, 2); logB = B.bit_length() - 1; lR = n_chunks.bit_length() - 1
        out = dict(R=np.zeros((n_chunks, 2), np.uint64), a=np.zeros((4, 2), np.uint64), poly=np.zeros((logB, 5, 2), np.uint64), gr=np.zeros((logB, 2), np.uint64),
                   fin6=np.zeros((6, 2), np.uint64), Peval=np.zeros((6, n_chunks, 2), np.uint64), b=np.zeros((6, 2), np.uint64), q2=np.zeros((lR, 3, 2), np.uint64),
                   r2=np.zeros((lR, 2), np.uint64), vr2=np.zeros((2, 2), np.uint64), fin2=np.zeros(2, np.uint64), checks=np.zeros(3, np.int32))
        names = ("R", "a", "poly", "gr", "fin6", "Peval", "b", "q2", "r2", "vr2", "fin2", "checks")

        class GO(ctypes.Structure):
            _fields_ = [(n, c_vp) for n in names]
        go = GO(*[out[k].ctypes.data for k in names])
        self._chk(self.lib.hobbit_gate_consistency_stream(self.ctx, ctypes.cast(tsrc, c_vp), None, n_chunks, B, _hp(r), ctypes.byref(go)))
        return out

    def set_lookups(self, lookup_rand):
        """the reference's has_lookups / lookup_rand globals (src/main.cpp:67,70); None switches the lookup gate maps off"""
        if lookup_rand is None:
            self._chk(self.lib.hobbit_set_lookups(self.ctx, 0, None))
        else:
            lr = Fh(lookup_rand).reshape(-1, 2)[:2].copy()
            self._chk(self.lib.hobbit_set_lookups(self.ctx, 1, _hp(lr)))

    def gate_consistency_lookups_stream(self, tsrc, n_chunks, B, r, lookup_rand):
        """prove_gate_consistency_lookups (src/sumcheck.cpp:503-795) over a trace source; sets has_lookups for the call and clears it after
        (lookup_rand None: leaves the flag as it is -- the library refuses to run with it unset)"""
        r = Fh(r).reshape(-1, 2); logB = B.bit_length() - 1; lR = n_chunks.bit_length() - 1
        out = dict(R=np.zeros((n_chunks, 2), np.uint64), a=np.zeros((5, 2), np.uint64), poly=np.zeros((logB, 5, 2), np.uint64), gr=np.zeros((logB, 2), np.uint64),
                   fin9=np.zeros((9, 2), np.uint64), Peval=np.zeros((8, n_chunks, 2), np.uint64), b=np.zeros((8, 2), np.uint64), q2=np.zeros((lR, 3, 2), np.uint64),
                   r2=np.zeros((lR, 2), np.uint64), vr2=np.zeros((2, 2), np.uint64), fin2=np.zeros(2, np.uint64), checks=np.zeros(5, np.int32))
        names = ("R", "a", "poly", "gr", "fin9", "Peval", "b", "q2", "r2", "vr2", "fin2", "checks")

        class GO(ctypes.Structure):
            _fields_ = [(n, c_vp) for n in names]
        go = GO(*[out[k].ctypes.data for k in names])
        if lookup_rand is not None:
            self.set_lookups(lookup_rand)
        try:
            self._chk(self.lib.hobbit_gate_consistency_lookups_stream(self.ctx, ctypes.cast(tsrc, c_vp), None, n_chunks, B, _hp(r), ctypes.byref(go)))
        finally:
            if lookup_rand is not None:
                self.set_lookups(None)
        return out

    def elastic_commit(self, N, B, opt, gcc_arg_order=1, chunk=None, keep_levels=False, levels="host"):
        """test_Elastic_PC's commit (src/Elastic_PC.cpp:736-771): opt 1 RSxRS trs=B/2^11, opt 2 RSxexpander trs=B/2^14.
        chunk: a DeviceBuffer holding the stream's (repeating) chunk, generated here on the host if None.
        levels="host": the whole tree comes back as a numpy array (what the reference's MT_hashes is; 64 B per leaf over PCIe);
        levels="device": returns (root, DeviceBuffer with the flat tree) -- the tree stays in HBM, as Our_PC's Commitment handle keeps it."""
        if opt == 1:
            lin, trs = 0, B >> 11
        else:
            lin, trs = 1, B >> 14
            self.expander_init_store(trs)
        e = c_vp()
        self._chk(self.lib.hobbit_elastic_begin(self.ctx, B, trs, lin, gcc_arg_order, ctypes.byref(e)))
        if chunk is None:
            chunk = self.to_device(self.read_stream_PC(B))
        for _ in range(N // B):
            self._chk(self.lib.hobbit_elastic_push(self.ctx, e, chunk.ptr))
        lv = self.alloc(32 * 8 * B)
        self._chk(self.lib.hobbit_elastic_finish(self.ctx, e, lv.ptr))
        if levels == "device":
            root = np.zeros(32, np.uint8)
            self._chk(self.lib.hobbit_memcpy_d2h(self.ctx, root.ctypes.data, lv.ptr + 32 * (8 * B - 2), 32))
            self.lib.hobbit_elastic_free(e)
            return root, lv
        out = self.to_host(lv, (8 * B - 1, 32), np.uint8)
        self.lib.hobbit_elastic_free(e)
        return (out, lv) if keep_levels else out

    # ---- sumchecks (reference names: src/sumcheck.cpp:2391, 1974)
    def _dev_table(self, v):
        if isinstance(v, tuple):
            return (v[0].ptr if isinstance(v[0], DeviceBuffer) else int(v[0])), v[1], None
        a = Fh(v).reshape(-1, 2)
        b = self.to_device(a)
        return b.ptr, a.shape[0], b

    def generate_2product_sumcheck_proof(self, v1, v2, previous_r):
        p1, n, k1 = self._dev_table(v1); p2, n2, k2 = self._dev_table(v2)
        assert n == n2
        rounds = n.bit_length() - 1
        pr = Fh(previous_r).reshape(2)
        q = np.zeros((rounds, 3, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64); vr = np.zeros((2, 2), np.uint64); fin = np.zeros(2, np.uint64)
        self._chk(self.lib.hobbit_sumcheck2(self.ctx, c_vp(p1), c_vp(p2), c_sz(n), _hp(pr), _hp(q), _hp(r), _hp(vr), _hp(fin)))
        return dict(poly=q, r=r, vr=vr, fin=fin)

    def generate_3product_sumcheck_proof(self, v1, v2, v3, previous_r):
        p1, n, k1 = self._dev_table(v1); p2, _, k2 = self._dev_table(v2); p3, _, k3 = self._dev_table(v3)
        rounds = n.bit_length() - 1
        pr = Fh(previous_r).reshape(2)
        q = np.zeros((rounds, 4, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64); vr = np.zeros((3, 2), np.uint64); fin = np.zeros(2, np.uint64)
        self._chk(self.lib.hobbit_sumcheck3(self.ctx, c_vp(p1), c_vp(p2), c_vp(p3), c_sz(n), _hp(pr), _hp(q), _hp(r), _hp(vr), _hp(fin)))
        return dict(poly=q, r=r, vr=vr, fin=fin)

    def prove_gate_consistency_standard(self, L, R, O, add, r):
        """src/sumcheck.cpp:434-501 (the in-memory gate-consistency prover): a = (1, 1, 1, -1), transcript seeded with F(213), claimed
        sum 0, mul_gate = 1 - add_gate, beta = precompute_beta(r).  The reference folds its inputs in place and returns nothing;
        returned here: element 0 of arr_L, arr_R, arr_O, add_gate after the last round."""
        L, R, O, add = [Fh(v).reshape(-1, 2) for v in (L, R, O, add)]
        P = np.uint64((1 << 61) - 1)
        mul = np.stack([(np.uint64(1) + P - add[:, 0]) % P, (P - add[:, 1]) % P], 1).astype(np.uint64)
        beta = self.precompute_beta(r)
        a = np.array([[1, 0], [1, 0], [1, 0], [int(P) - 1, 0]], np.uint64)
        res = self.gate_sumcheck((add, beta, L, R, O, mul), a, np.array([213, 0], np.uint64), np.array([0, 0], np.uint64))
        return np.stack([res["fin"][2], res["fin"][3], res["fin"][4], res["fin"][0]])

    gate_standard = prove_gate_consistency_standard

    def gate_sumcheck(self, tables, a, rand, claimed_sum):
        """degree-4 gate-consistency sumcheck (src/sumcheck.cpp:875-929); tables = (add, beta, L, R, O, mul)"""
        devs = [self._dev_table(t) for t in tables]
        n = devs[0][1]; rounds = n.bit_length() - 1
        a = Fh(a).reshape(4, 2); rnd = Fh(rand).reshape(2).copy(); sm = Fh(claimed_sum).reshape(2).copy()
        q = np.zeros((rounds, 5, 2), np.uint64); r = np.zeros((rounds, 2), np.uint64); fin = np.zeros((6, 2), np.uint64); chk = ctypes.c_int(0)
        self._chk(self.lib.hobbit_gate_sumcheck(self.ctx, *[c_vp(d[0]) for d in devs], c_sz(n), _hp(a), _hp(rnd), _hp(sm), _hp(q), _hp(r), _hp(fin), ctypes.byref(chk)))
        return dict(poly=q, r=r, fin=fin, rand=rnd, sum=sm, check=np.array([chk.value], np.int32))


def splitmix_field(n, seed):
    """n full-range F_{p^2} elements (splitmix64 mod p): the synthetic-input generator of SURVEY.md 8(d), host side
    (hobbit_fill_splitmix is its device twin)."""
    P = (1 << 61) - 1
    idx = np.arange(1, 2 * n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * np.uint64(0x632BE59BD9B4E019) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z % np.uint64(P)).reshape(n, 2)


# ---- submodule: chunk-sharded multi-GPU commit orchestration --------------------------------
def _load_parallel():
    name = __name__ + ".parallel"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_DIR, "parallel.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


parallel = _load_parallel()
parallel_tree_top = parallel.tree_top_host
