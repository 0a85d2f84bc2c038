"""ctypes binding of libdapol_hip.so (include/dapol_hip.h).  Plumbing only: numpy arrays in, numpy arrays out.

The library is the product; this module never computes anything itself and raises if the HIP library is
missing -- there is no CPU fallback."""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DAPOL_HIP_LIB", os.path.join(_HERE, "libdapol_hip.so"))   # override only for A/B builds

POLICY_PADDING, POLICY_SPLITTING = 0, 1
PROFILE_BENCH, PROFILE_HOST = 0, 1          # dapol_options.profile: memory defaults tuned for the bench / for an embedder that shares the GPU
DIGEST_BLAKE3, DIGEST_BLAKE2S, DIGEST_BLAKE2B = 0, 1, 2      # Blake2b: 64-byte node hashes (every H array is then (.., 64))


class DapolError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dapol_hip status {code}: {msg}")
        self.code = code


class WorkloadStats(ctypes.Structure):
    _fields_ = [("tree_ms", ctypes.c_double), ("prove_ms", ctypes.c_double), ("msm_ms", ctypes.c_double),
                ("msm_launches", ctypes.c_uint64), ("proofs", ctypes.c_uint64), ("proof_bytes", ctypes.c_uint64),
                ("checksum", ctypes.c_uint64), ("root_C", ctypes.c_uint8 * 32), ("root_H", ctypes.c_uint8 * 32),
                ("mat_ms", ctypes.c_double), ("mat_launches", ctypes.c_uint64), ("msm_kernels", ctypes.c_uint64), ("mat_kernels", ctypes.c_uint64),
                ("msm_all_ms", ctypes.c_double), ("msm_span_ms", ctypes.c_double)]


class CommTiming(ctypes.Structure):
    """dapol_comm_timing (include/dapol_hip.h): microseconds, measured inside dapol_shard_exchange / dapol_comm_allreduce_u64."""
    _fields_ = [("exchanges", ctypes.c_uint64), ("reduces", ctypes.c_uint64)] + [(k, ctypes.c_double) for k in (
        "last_allgather_us", "last_top_levels_us", "last_exchange_host_us", "last_allreduce_us", "last_reduce_host_us",
        "sum_allgather_us", "sum_top_levels_us", "sum_exchange_host_us", "sum_allreduce_us", "sum_reduce_host_us")]


class Options(ctypes.Structure):
    """dapol_options (include/dapol_hip.h): zeros = the library's own choices."""
    _fields_ = [("struct_size", ctypes.c_int32), ("window_bits", ctypes.c_int32), ("table_gb", ctypes.c_double), ("high_half_rows", ctypes.c_int32),
                ("generator_stationary", ctypes.c_int32), ("gs_tile_rows", ctypes.c_int32), ("streams", ctypes.c_int32), ("chunk_proofs", ctypes.c_int64),
                ("scratch_gb", ctypes.c_double), ("tail_length", ctypes.c_int32), ("small_call_max", ctypes.c_int32), ("verify_batch_min", ctypes.c_int32),
                ("update_incremental_max", ctypes.c_int64), ("gs_slices", ctypes.c_int32), ("profile", ctypes.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        self.struct_size = ctypes.sizeof(Options)
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError("dapol_options has no field " + k)
            setattr(self, k, v)


_lib = None

_P = ctypes.c_void_p
_SIG = {
    "dapol_ctx_create": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_P)]),
    "dapol_ctx_create_opts": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(Options), ctypes.POINTER(_P)]),
    "dapol_ctx_get_options": (ctypes.c_int32, [_P, ctypes.POINTER(Options)]),
    "dapol_ctx_set_options": (ctypes.c_int32, [_P, ctypes.POINTER(Options)]),
    "dapol_env_knobs": (ctypes.c_int32, [ctypes.c_int32]),
    "dapol_ctx_destroy": (ctypes.c_int32, [_P]),
    "dapol_ctx_digest_bytes": (ctypes.c_int32, [_P, _P]),
    "dapol_proof_nodes_serialize_d": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_size_t, _P, _P, _P]),
    "dapol_proof_wire_size_d": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dapol_proof_serialize_d": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, _P, ctypes.c_size_t, _P, _P, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.c_int32, _P, _P]),
    "dapol_ctx_generator": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P]),
    "dapol_strerror": (ctypes.c_char_p, [ctypes.c_int32]),
    "dapol_last_error": (ctypes.c_char_p, []),
    "dapol_commit_hash_batch": (ctypes.c_int32, [_P, ctypes.c_size_t, _P, _P, _P, _P]),
    "dapol_build_leaf_nodes": (ctypes.c_int32, [_P, ctypes.c_int32, _P, ctypes.c_size_t, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, _P, _P, _P, _P, _P,
                                                _P, _P]),
    "dapol_tree_build": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, _P, ctypes.c_int32, ctypes.POINTER(_P)]),
    "dapol_tree_padding_positions": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, _P]),
    "dapol_tree_build_tape": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, _P, ctypes.c_size_t, ctypes.POINTER(_P)]),
    "dapol_entity_tape_slots": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dapol_prove_entities_tape": (ctypes.c_int32, [_P, _P, ctypes.c_size_t, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P, _P, _P, _P]),
    "dapol_tree_build_shard": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, _P, ctypes.POINTER(_P)]),
    "dapol_merge_batch": (ctypes.c_int32, [_P, ctypes.c_size_t] + [_P] * 12),
    "dapol_padding_nodes": (ctypes.c_int32, [_P, _P, ctypes.c_size_t, _P, _P, _P, _P, _P]),
    "dapol_tree_destroy": (ctypes.c_int32, [_P]),
    "dapol_tree_root": (ctypes.c_int32, [_P, _P, _P, _P, _P]),
    "dapol_tree_node_count": (ctypes.c_int32, [_P, _P, _P]),
    "dapol_tree_update": (ctypes.c_int32, [_P, ctypes.c_size_t, _P, _P, _P]),
    "dapol_tree_last_update_path": (ctypes.c_int32, [_P, _P]),
    "dapol_diag_fork_guard_waits": (ctypes.c_int32, [_P]),
    "dapol_diag_verify_fallbacks": (ctypes.c_int32, [_P]),
    "dapol_diag_range_prove_ms": (ctypes.c_int32, [_P]),
    "dapol_tree_level_size": (ctypes.c_int32, [_P, ctypes.c_int32, _P, _P]),
    "dapol_tree_level_nodes": (ctypes.c_int32, [_P, ctypes.c_int32, _P, _P, _P, _P, _P, _P]),
    "dapol_tree_paths": (ctypes.c_int32, [_P, ctypes.c_size_t, _P, _P, _P, _P, _P]),
    "dapol_range_prove_batch": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, _P, ctypes.c_uint64, _P, _P]),
    "dapol_range_proof_size": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "dapol_range_verify_batch": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, _P]),
    "dapol_prove_entities": (ctypes.c_int32, [_P, _P, ctypes.c_size_t, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P, _P, _P, _P]),
    "dapol_range_proofs_wire_size": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dapol_range_proofs_serialize": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P, _P]),
    "dapol_range_proofs_deserialize": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_size_t, _P, ctypes.c_size_t, _P, _P, _P, _P]),
    "dapol_verify_entities": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_int32, _P, _P, _P]),
    "dapol_verify_entities_checked": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, ctypes.c_size_t, _P, _P, _P, _P, ctypes.c_int32,
                                                       ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_size_t, _P, _P]),
    "dapol_verify_batch_checked": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, ctypes.c_size_t, _P, _P, _P, _P, ctypes.c_int32,
                                                    ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_size_t, _P, _P]),
    "dapol_batch_siblings": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, _P]),
    "dapol_prove_batch": (ctypes.c_int32, [_P, _P, ctypes.c_size_t, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P, _P, _P, _P]),
    "dapol_verify_batch": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, ctypes.c_size_t, _P, _P, _P, _P, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_int32, _P, _P, _P]),
    "dapol_entity_proof_size": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dapol_prove_entities_upper": (ctypes.c_int32, [_P, _P, ctypes.c_size_t, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_int32,
                                                    _P, _P, _P, _P, _P, _P, _P]),
    "dapol_workload_create": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, ctypes.POINTER(_P)]),
    "dapol_workload_create_shard": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, _P, _P, _P, ctypes.POINTER(_P)]),
    "dapol_workload_build": (ctypes.c_int32, [_P, _P, _P, _P, _P, _P, ctypes.POINTER(WorkloadStats)]),
    "dapol_workload_prove": (ctypes.c_int32, [_P, _P, ctypes.c_int32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int32, _P, _P, _P, _P,
                                              ctypes.POINTER(WorkloadStats)]),
    "dapol_workload_prove_policy": (ctypes.c_int32, [_P, _P, ctypes.c_int32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                     _P, _P, _P, _P, ctypes.POINTER(WorkloadStats)]),
    "dapol_workload_paths": (ctypes.c_int32, [_P, ctypes.c_size_t, _P, ctypes.c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dapol_workload_destroy": (ctypes.c_int32, [_P]),
    "dapol_workload_run": (ctypes.c_int32, [_P, _P, _P, ctypes.c_int32, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(WorkloadStats)]),
    "dapol_workload_proofs": (ctypes.c_int32, [_P, ctypes.c_size_t, ctypes.c_size_t, _P]),
    "dapol_tree_node_records": (ctypes.c_int32, [_P, ctypes.c_size_t, _P, _P, _P, _P, _P, _P, _P]),
    "dapol_shard_top_node_records": (ctypes.c_int32, [_P, ctypes.c_int32, _P, ctypes.c_size_t, _P, _P, _P, _P, _P, _P]),
    "dapol_prove_batch_records": (ctypes.c_int32, [_P, ctypes.c_size_t, _P, ctypes.c_size_t, _P, _P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                   _P, _P]),
    "dapol_workload_tree": (ctypes.c_int32, [_P, ctypes.POINTER(_P)]),
    "dapol_comm_unique_id": (ctypes.c_int32, [_P]),
    "dapol_comm_create": (ctypes.c_int32, [_P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_P)]),
    "dapol_comm_create_timeout": (ctypes.c_int32, [_P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(_P)]),
    "dapol_comm_destroy": (ctypes.c_int32, [_P]),
    "dapol_comm_abort": (ctypes.c_int32, [_P]),
    "dapol_comm_count": (ctypes.c_int32, [_P, _P]),
    "dapol_shard_exchange": (ctypes.c_int32, [_P, _P, _P, ctypes.c_uint64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dapol_shard_top_levels": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dapol_comm_allreduce_u64": (ctypes.c_int32, [_P, ctypes.c_int32, _P, ctypes.c_size_t]),
    "dapol_comm_timing_get": (ctypes.c_int32, [_P, _P, ctypes.c_int32]),
    "dapol_wire_config_get": (ctypes.c_int32, [_P]),
    "dapol_wire_config_set": (ctypes.c_int32, [_P]),
    "dapol_proof_nodes_serialize": (ctypes.c_int32, [ctypes.c_size_t, _P, _P, _P]),
    "dapol_proof_nodes_deserialize": (ctypes.c_int32, [_P, ctypes.c_size_t, _P, ctypes.c_size_t, _P, _P]),
    "dapol_proof_wire_size": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "dapol_proof_serialize": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_size_t, _P, ctypes.c_size_t, _P, _P, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.c_int32, _P, _P]),
    "dapol_proof_deserialize": (ctypes.c_int32, [_P, ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_size_t, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
}
EXPORTED_SYMBOLS = sorted(_SIG)


def lib():
    """Loads libdapol_hip.so (built by __graft_entry__.build()).  Raises if it is missing: no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DapolError(-1, f"{LIB_PATH} not found: run `python __graft_entry__.py` (hipcc) first; there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIG.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _chk(code):
    if code != 0:
        L = lib()
        raise DapolError(code, f"{L.dapol_strerror(code).decode()} ({L.dapol_last_error().decode()})")


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _u8(a, *shape):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if shape:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_P)


def range_proofs_serialize(height, policy, aggregation_factor, n_bits, blob):
    """R::serialize() of one entity's proofs (host-only)."""
    blob = _u8(np.frombuffer(bytes(blob), np.uint8))
    n = lib().dapol_range_proofs_wire_size(height, policy, aggregation_factor, n_bits)
    out = np.zeros(max(n, 1), np.uint8)
    _chk(lib().dapol_range_proofs_serialize(height, policy, aggregation_factor, n_bits, _ptr(blob), _ptr(out)))
    return out[:n].tobytes()


def range_proofs_deserialize(policy, n_bits, wire):
    """R::deserialize(): returns (aggregated [bytes], individual [bytes], consumed)."""
    w = _u8(np.frombuffer(bytes(wire), np.uint8)) if len(wire) else np.zeros(1, np.uint8)
    blob = np.zeros(len(wire) + 1, np.uint8)
    nagg, nind, cons = ctypes.c_uint32(), ctypes.c_uint64(), ctypes.c_size_t()
    sizes = np.zeros(16, np.uint64)
    _chk(lib().dapol_range_proofs_deserialize(policy, n_bits, _ptr(w), len(wire), _ptr(blob), len(blob), ctypes.byref(nagg), _ptr(sizes),
                                              ctypes.byref(nind), ctypes.byref(cons)))
    b, pos, agg = blob.tobytes(), 0, []
    for i in range(nagg.value):
        agg.append(b[pos:pos + int(sizes[i])])
        pos += int(sizes[i])
    ps1 = lib().dapol_range_proof_size(n_bits, 1)
    ind = [b[pos + i * ps1:pos + (i + 1) * ps1] for i in range(nind.value)]
    return agg, ind, cons.value


class WireConfig(ctypes.Structure):
    """dapol_wire_config: the smtree assumptions (byte order, field widths, sibling order), one field each."""
    _fields_ = [("int_big_endian", ctypes.c_int32), ("batch_num_bytes", ctypes.c_int32), ("sibling_num_bytes", ctypes.c_int32),
                ("tree_height_bytes", ctypes.c_int32), ("path_bytes_full", ctypes.c_int32), ("siblings_leaf_first", ctypes.c_int32)]


def wire_config_get():
    c = WireConfig()
    _chk(lib().dapol_wire_config_get(ctypes.byref(c)))
    return c


def wire_config_set(**fields):
    """Updates the given fields of the process-wide dapol_wire_config; returns the previous config (restore with
    wire_config_restore)."""
    old = wire_config_get()
    new = WireConfig.from_buffer_copy(bytes(old))
    for k, v in fields.items():
        setattr(new, k, int(v))
    _chk(lib().dapol_wire_config_set(ctypes.byref(new)))
    return old


def wire_config_restore(cfg):
    _chk(lib().dapol_wire_config_set(ctypes.byref(cfg)))


def proof_serialize(height, leaf_idx, sib_C, sib_H, policy, aggregation_factor, n_bits, range_blob, hash_bytes=32):
    """DapolProof::serialize (range_proof || merkle_path) for a proof over the given leaves and siblings (host-only).
    hash_bytes = D::output_size() of the tree's digest (64 for a Blake2b context)."""
    leaf_idx = _u64(leaf_idx)
    sC, sH = _u8(sib_C).reshape(-1, 32), _u8(sib_H).reshape(-1, hash_bytes)
    S = sC.shape[0]
    n = lib().dapol_proof_wire_size_d(hash_bytes, height, leaf_idx.shape[0], S, policy, aggregation_factor, n_bits)
    if n == 0:
        raise DapolError(8, "bad policy / aggregation_factor / n_bits")
    blob = _u8(np.frombuffer(bytes(range_blob), np.uint8))
    out = np.zeros(n, np.uint8)
    _chk(lib().dapol_proof_serialize_d(hash_bytes, height, leaf_idx.shape[0], _ptr(leaf_idx), S, _ptr(sC) if S else None, _ptr(sH) if S else None, policy,
                                       aggregation_factor, n_bits, _ptr(blob), _ptr(out)))
    return out.tobytes()


class Context:
    """dapol_ctx: generators + window tables on one GPU."""

    def proof_deserialize(self, policy, n_bits, wire):
        """DapolProof::deserialize -> dict(height, leaf_idx, sib_C, sib_H, aggregation_factor, range_blob, consumed); the sibling
        commitments are decompress-validated on the GPU (DapolError 6 / 7 = BytesNotEnough / ValueDecodingError)."""
        w = _u8(np.frombuffer(bytes(wire), np.uint8)) if len(wire) else np.zeros(1, np.uint8)
        h, agg = ctypes.c_int32(), ctypes.c_int32()
        k, S, bl, cons = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        args = (self.h, policy, n_bits, _ptr(w), len(wire), ctypes.byref(h), ctypes.byref(k), ctypes.byref(S), ctypes.byref(agg), ctypes.byref(bl))
        _chk(lib().dapol_proof_deserialize(*args, None, None, None, None, ctypes.byref(cons)))
        leaf = np.zeros(max(k.value, 1), np.uint64)
        sC, sH = np.zeros((max(S.value, 1), 32), np.uint8), np.zeros((max(S.value, 1), self.hb), np.uint8)
        blob = np.zeros(max(bl.value, 1), np.uint8)
        _chk(lib().dapol_proof_deserialize(*args, _ptr(leaf), _ptr(sC), _ptr(sH), _ptr(blob), ctypes.byref(cons)))
        return dict(height=h.value, leaf_idx=leaf[:k.value], sib_C=sC[:S.value], sib_H=sH[:S.value], aggregation_factor=agg.value,
                    range_blob=blob[:bl.value].tobytes(), consumed=cons.value)

    def proof_nodes_deserialize(self, wire, n):
        w = _u8(np.frombuffer(bytes(wire), np.uint8)) if len(wire) else np.zeros(1, np.uint8)
        C, H = np.zeros((max(n, 1), 32), np.uint8), np.zeros((max(n, 1), self.hb), np.uint8)
        _chk(lib().dapol_proof_nodes_deserialize(self.h, n, _ptr(w), len(wire), _ptr(C), _ptr(H)))
        return C[:n], H[:n]

    def __init__(self, device=0, max_parties=32, digest=DIGEST_BLAKE3, options=None):
        """digest: the node hash D of Dapol<D, R> (DIGEST_BLAKE3 or DIGEST_BLAKE2S); options: an Options (dapol_options)."""
        self.h = _P()
        if options is None:
            _chk(lib().dapol_ctx_create(device, max_parties, digest, ctypes.byref(self.h)))
        else:
            _chk(lib().dapol_ctx_create_opts(device, max_parties, digest, ctypes.byref(options), ctypes.byref(self.h)))
        self.max_parties = max_parties
        hb = ctypes.c_int32(0)
        _chk(lib().dapol_ctx_digest_bytes(self.h, ctypes.byref(hb)))
        self.hb = int(hb.value)                   # bytes of one node hash: the last dimension of every H array of this context

    def close(self):
        if self.h:
            lib().dapol_ctx_destroy(self.h)
            self.h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def generator(self, which, party=0, bit=0):
        out = np.zeros(32, np.uint8)
        _chk(lib().dapol_ctx_generator(self.h, which, party, bit, _ptr(out)))
        return out.tobytes()

    def commit_hash_batch(self, v, r32):
        v = _u64(v)
        n = v.shape[0]
        r32 = _u8(r32, n, 32)
        C, H = np.zeros((n, 32), np.uint8), np.zeros((n, self.hb), np.uint8)
        _chk(lib().dapol_commit_hash_batch(self.h, n, _ptr(v), _ptr(r32), _ptr(C), _ptr(H)))
        return C, H

    def build_leaf_nodes(self, liabilities, audit_seed, height, digest=DIGEST_BLAKE3):
        """build_leaf_nodes (src/dapol/mod.rs:323-399): liabilities = [(internal_id bytes, external_id bytes, value)] in
        input order -> dict(leaf_idx, v, r (sorted by index), order (input position per sorted leaf), idx_by_entity)."""
        n = len(liabilities)
        iid = b"".join(l[0] for l in liabilities)
        eid = b"".join(l[1] for l in liabilities)
        ioff = np.zeros(n + 1, np.uint32)
        eoff = np.zeros(n + 1, np.uint32)
        ioff[1:] = np.cumsum([len(l[0]) for l in liabilities])
        eoff[1:] = np.cumsum([len(l[1]) for l in liabilities])
        vals = _u64([l[2] for l in liabilities])
        ib = _u8(np.frombuffer(iid, np.uint8)) if iid else np.zeros(1, np.uint8)
        eb = _u8(np.frombuffer(eid, np.uint8)) if eid else np.zeros(1, np.uint8)
        sd = _u8(np.frombuffer(audit_seed, np.uint8)) if audit_seed else None
        idx, v, order, by_e = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint64)
        r = np.zeros((n, 32), np.uint8)
        _chk(lib().dapol_build_leaf_nodes(self.h, digest, _ptr(sd), len(audit_seed), height, n, _ptr(ib), _ptr(ioff), _ptr(eb), _ptr(eoff),
                                          _ptr(vals), _ptr(idx), _ptr(v), _ptr(r), _ptr(order), _ptr(by_e)))
        return dict(leaf_idx=idx, v=v, r=r, order=order, idx_by_entity=by_e)

    def get_options(self):
        o = Options()
        _chk(lib().dapol_ctx_get_options(self.h, ctypes.byref(o)))
        return o

    def set_options(self, options):
        _chk(lib().dapol_ctx_set_options(self.h, ctypes.byref(options)))

    def build_leaf_nodes_packed(self, iid, ioff, eid, eoff, values, audit_seed, height, digest=DIGEST_BLAKE3):
        """build_leaf_nodes over PACKED ids (the C ABI's own layout: id i = iid[ioff[i]:ioff[i + 1]]) -- what a caller with a
        million liabilities hands over; same result dict as build_leaf_nodes."""
        ioff, eoff = np.ascontiguousarray(ioff, np.uint32), np.ascontiguousarray(eoff, np.uint32)
        n = ioff.shape[0] - 1
        vals = _u64(values)
        ib = _u8(np.frombuffer(bytes(iid), np.uint8)) if len(iid) else np.zeros(1, np.uint8)
        eb = _u8(np.frombuffer(bytes(eid), np.uint8)) if len(eid) else np.zeros(1, np.uint8)
        sd = _u8(np.frombuffer(audit_seed, np.uint8)) if audit_seed else None
        idx, v, order, by_e = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.uint64)
        r = np.zeros((n, 32), np.uint8)
        _chk(lib().dapol_build_leaf_nodes(self.h, digest, _ptr(sd), len(audit_seed), height, n, _ptr(ib), _ptr(ioff), _ptr(eb), _ptr(eoff),
                                          _ptr(vals), _ptr(idx), _ptr(v), _ptr(r), _ptr(order), _ptr(by_e)))
        return dict(leaf_idx=idx, v=v, r=r, order=order, idx_by_entity=by_e)

    def padding_nodes(self, pad_seed, level, index):
        """Paddable::padding at the given (level above the leaves, index) positions: (C, H, r); the value is 0."""
        level, index = _u8(level), _u64(index)
        n = index.shape[0]
        C, H, r = np.zeros((n, 32), np.uint8), np.zeros((n, self.hb), np.uint8), np.zeros((n, 32), np.uint8)
        seed = _u8(np.frombuffer(pad_seed, np.uint8))
        _chk(lib().dapol_padding_nodes(self.h, _ptr(seed), n, _ptr(level), _ptr(index), _ptr(C), _ptr(H), _ptr(r)))
        return C, H, r

    def merge_batch(self, CL, HL, CR, HR, vL=None, rL=None, vR=None, rR=None):
        """Mergeable::merge on compressed records; returns (C, H) or (C, H, v, r) when the secrets are given."""
        CL = _u8(CL).reshape(-1, 32)
        n = CL.shape[0]
        HL, CR, HR = _u8(HL, n, self.hb), _u8(CR, n, 32), _u8(HR, n, self.hb)
        C, H = np.zeros((n, 32), np.uint8), np.zeros((n, self.hb), np.uint8)
        if vL is None:
            _chk(lib().dapol_merge_batch(self.h, n, _ptr(CL), _ptr(HL), None, None, _ptr(CR), _ptr(HR), None, None, _ptr(C), _ptr(H), None, None))
            return C, H
        vL, vR, rL, rR = _u64(vL), _u64(vR), _u8(rL, n, 32), _u8(rR, n, 32)
        v, r = np.zeros(n, np.uint64), np.zeros((n, 32), np.uint8)
        _chk(lib().dapol_merge_batch(self.h, n, _ptr(CL), _ptr(HL), _ptr(vL), _ptr(rL), _ptr(CR), _ptr(HR), _ptr(vR), _ptr(rR), _ptr(C), _ptr(H),
                                     _ptr(v), _ptr(r)))
        return C, H, v, r

    def range_prove_batch(self, n_bits, m, v, r32, nonce_seed=None, stream_id=None, slot_base=0, tape=None):
        v = _u64(v).reshape(-1, m)
        b = v.shape[0]
        r32 = _u8(r32, b, m, 32)
        ps = lib().dapol_range_proof_size(n_bits, m)
        out = np.zeros((b, max(ps, 1)), np.uint8)
        seed = _u8(np.frombuffer(nonce_seed, np.uint8)) if nonce_seed is not None else None
        sid = _u64(stream_id) if stream_id is not None else None
        tp = _u8(tape) if tape is not None else None
        _chk(lib().dapol_range_prove_batch(self.h, n_bits, m, b, _ptr(v), _ptr(r32), _ptr(seed), _ptr(sid), slot_base, _ptr(tp), _ptr(out)))
        return out

    def verify_entities(self, height, leaf_idx, leaf_C, leaf_H, path_C, path_H, root_C, root_H, policy, aggregation_factor, n_bits, range_proofs,
                        verify_seed=None):
        """DapolProof::verify for single-leaf proofs (Merkle re-merge + policy range verification)."""
        leaf_idx = _u64(leaf_idx)
        b = leaf_idx.shape[0]
        lC, lH = _u8(leaf_C, b, 32), _u8(leaf_H, b, self.hb)
        # the length-checked entry point: arrays whose sizes do not fit (height, policy, aggregation factor) -- proofs decoded from
        # hostile bytes -- come back as invalid instead of being over-read
        pC, pH = _u8(path_C).reshape(-1, 32), _u8(path_H).reshape(-1, self.hb)
        rp = _u8(range_proofs).reshape(-1)
        rC, rH = _u8(np.frombuffer(root_C, np.uint8)), _u8(np.frombuffer(root_H, np.uint8))
        seed = _u8(np.frombuffer(verify_seed, np.uint8)) if verify_seed is not None else None   # None: the library draws one from the OS
        ok = np.zeros(b, np.uint8)
        if pH.shape[0] != pC.shape[0]:
            return ok
        _chk(lib().dapol_verify_entities_checked(self.h, height, b, _ptr(leaf_idx), _ptr(lC), _ptr(lH), pC.shape[0], _ptr(pC), _ptr(pH), _ptr(rC), _ptr(rH),
                                                 policy, aggregation_factor, n_bits, _ptr(rp), rp.shape[0], _ptr(seed), _ptr(ok)))
        return ok

    def verify_batch(self, height, leaf_idx, leaf_C, leaf_H, sib_C, sib_H, root_C, root_H, policy, aggregation_factor, n_bits, range_proofs,
                     verify_seed=None):
        """DapolProof::verify_batch: one proof covering k leaves."""
        leaf_idx = _u64(leaf_idx)
        k = leaf_idx.shape[0]
        lC, lH = _u8(leaf_C, k, 32), _u8(leaf_H, k, self.hb)
        sC, sH = _u8(sib_C).reshape(-1, 32), _u8(sib_H).reshape(-1, self.hb)
        rp = _u8(np.frombuffer(bytes(range_proofs), np.uint8))
        rC, rH = _u8(np.frombuffer(root_C, np.uint8)), _u8(np.frombuffer(root_H, np.uint8))
        seed = _u8(np.frombuffer(verify_seed, np.uint8)) if verify_seed is not None else None   # None: the library draws one from the OS
        ok = np.zeros(1, np.uint8)
        if sH.shape[0] != sC.shape[0]:
            return False
        _chk(lib().dapol_verify_batch_checked(self.h, height, k, _ptr(leaf_idx), _ptr(lC), _ptr(lH), sC.shape[0], _ptr(sC), _ptr(sH), _ptr(rC), _ptr(rH),
                                              policy, aggregation_factor, n_bits, _ptr(rp), rp.shape[0], _ptr(seed), _ptr(ok)))
        return bool(ok[0])

    def range_verify_batch(self, n_bits, m, proofs, V32, verify_seed=None):
        proofs = _u8(proofs)
        b = proofs.shape[0]
        V32 = _u8(V32, b, m, 32)
        ok = np.zeros(b, np.uint8)
        seed = _u8(np.frombuffer(verify_seed, np.uint8)) if verify_seed is not None else None   # None: the library draws one from the OS
        _chk(lib().dapol_range_verify_batch(self.h, n_bits, m, b, _ptr(proofs), _ptr(V32), _ptr(seed), _ptr(ok)))
        return ok


def tree_padding_positions(height, leaf_idx):
    """(level, index) of the padding nodes of the tree over the given leaves, in tape order (level bottom-up, index ascending)."""
    leaf_idx = _u64(leaf_idx)
    n = ctypes.c_size_t(0)
    _chk(lib().dapol_tree_padding_positions(height, leaf_idx.shape[0], _ptr(leaf_idx), ctypes.byref(n), None, None))
    level, index = np.zeros(max(n.value, 1), np.uint8), np.zeros(max(n.value, 1), np.uint64)
    _chk(lib().dapol_tree_padding_positions(height, leaf_idx.shape[0], _ptr(leaf_idx), ctypes.byref(n), _ptr(level), _ptr(index)))
    return level[:n.value], index[:n.value]


def batch_siblings(height, leaf_idx):
    """Positions (level, index) of the siblings of a batched Merkle proof, in proof order (host-only)."""
    leaf_idx = _u64(leaf_idx)
    n = ctypes.c_size_t(0)
    _chk(lib().dapol_batch_siblings(height, leaf_idx.shape[0], _ptr(leaf_idx), ctypes.byref(n), None, None))
    level, index = np.zeros(n.value, np.uint8), np.zeros(n.value, np.uint64)
    _chk(lib().dapol_batch_siblings(height, leaf_idx.shape[0], _ptr(leaf_idx), ctypes.byref(n), _ptr(level), _ptr(index)))
    return level, index


COMM_ID_BYTES, RECORD_BYTES = 128, 104
REDUCE_SUM, REDUCE_MIN, REDUCE_MAX = 0, 1, 2


def comm_unique_id():
    """ncclGetUniqueId through the C ABI (rank 0 calls it; the host distributes the 128 bytes)."""
    out = np.zeros(COMM_ID_BYTES, np.uint8)
    _chk(lib().dapol_comm_unique_id(_ptr(out)))
    return out.tobytes()


def _top_out(bits):
    return (np.zeros(32, np.uint8), np.zeros(32, np.uint8), np.zeros(1, np.uint64), np.zeros(32, np.uint8),
            np.zeros((max(bits, 1), 32), np.uint8), np.zeros((max(bits, 1), 32), np.uint8), np.zeros(max(bits, 1), np.uint64),
            np.zeros((max(bits, 1), 32), np.uint8))


def _top_result(o, bits):
    rC, rH, rv, rr, uC, uH, uv, ur = o
    return (rC.tobytes(), rH.tobytes(), int(rv[0]), rr.tobytes()), (uC[:bits], uH[:bits], uv[:bits], ur[:bits])


class Comm:
    """dapol_comm: the RCCL communicator of the sharded path (one rank per GPU), created inside the library."""

    def __init__(self, ctx, unique_id, rank, world, timeout_s=0.0):
        """timeout_s > 0: the communicator is aborted (ncclCommAbort) and DapolError raised if it has not come up by then."""
        self.ctx, self.rank, self.world = ctx, rank, world
        self.bits = world.bit_length() - 1
        uid = _u8(np.frombuffer(unique_id, np.uint8))
        self.h = _P()
        _chk(lib().dapol_comm_create_timeout(ctx.h, _ptr(uid), rank, world, int(timeout_s * 1000), ctypes.byref(self.h)))

    def close(self):
        if self.h:
            lib().dapol_comm_destroy(self.h)
            self.h = _P()

    def abort(self):
        """The failure path: ncclCommAbort, without waiting for the peers."""
        if self.h:
            lib().dapol_comm_abort(self.h)
            self.h = _P()

    def count(self):
        n = ctypes.c_int32(0)
        _chk(lib().dapol_comm_count(self.h, ctypes.byref(n)))
        return int(n.value)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def exchange(self, root, with_records=False):
        """ncclAllGather of the subtree-root records + the replicated top levels: (global root record, upper siblings)."""
        C, H, v, r = root
        o = _top_out(self.bits)
        rec = np.zeros((self.world, RECORD_BYTES), np.uint8) if with_records else None
        _chk(lib().dapol_shard_exchange(self.h, _ptr(_u8(np.frombuffer(C, np.uint8))), _ptr(_u8(np.frombuffer(H, np.uint8))), int(v),
                                        _ptr(_u8(np.frombuffer(r, np.uint8))), *[_ptr(x) for x in o], _ptr(rec)))
        res = _top_result(o, self.bits)
        return res + (rec,) if with_records else res

    def allreduce(self, words, op=REDUCE_SUM):
        w = _u64(words).copy()
        _chk(lib().dapol_comm_allreduce_u64(self.h, op, _ptr(w), w.shape[0]))
        return w

    def timing(self, reset=False):
        """dapol_comm_timing_get: device / host microseconds of the collectives made through this communicator."""
        t = CommTiming()
        _chk(lib().dapol_comm_timing_get(self.h, ctypes.byref(t), 1 if reset else 0))
        return t


def tree_node_records(tree_handle, level, index):
    """Records (C, H, v, r, found) of the nodes a (shard) tree stores at the given (level above the leaves, index) positions."""
    level, index = _u8(level), _u64(index)
    n = index.shape[0]
    C, H, r = (np.zeros((max(n, 1), 32), np.uint8) for _ in range(3))
    v, found = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint8)
    _chk(lib().dapol_tree_node_records(tree_handle, n, _ptr(level), _ptr(index), _ptr(C), _ptr(H), _ptr(v), _ptr(r), _ptr(found)))
    return C[:n], H[:n], v[:n], r[:n], found[:n]


def shard_top_node_records(ctx, records, level_above, index):
    rec = _u8(records).reshape(-1, RECORD_BYTES)
    level_above, index = _u8(level_above), _u64(index)
    n = index.shape[0]
    C, H, r = (np.zeros((max(n, 1), 32), np.uint8) for _ in range(3))
    v = np.zeros(max(n, 1), np.uint64)
    _chk(lib().dapol_shard_top_node_records(ctx.h, rec.shape[0], _ptr(rec), n, _ptr(level_above), _ptr(index), _ptr(C), _ptr(H), _ptr(v), _ptr(r)))
    return C[:n], H[:n], v[:n], r[:n]


def prove_batch_records(ctx, leaf_idx, sib_C, sib_v, sib_r, policy, aggregation_factor, n_bits, nonce_seed):
    """R::generate_proof over assembled sibling records (dapol_batch_siblings order): the range-proof blob."""
    leaf_idx = _u64(leaf_idx)
    sC, sr, sv = _u8(sib_C).reshape(-1, 32), _u8(sib_r).reshape(-1, 32), _u64(sib_v)
    S = sC.shape[0]
    es = lib().dapol_entity_proof_size(S, policy, aggregation_factor, n_bits)
    if es == 0:
        raise DapolError(8, "bad policy / aggregation_factor / n_bits")
    out = np.zeros(es, np.uint8)
    seed = _u8(np.frombuffer(nonce_seed, np.uint8))
    _chk(lib().dapol_prove_batch_records(ctx.h, leaf_idx.shape[0], _ptr(leaf_idx), S, _ptr(sC), _ptr(sv), _ptr(sr), policy, aggregation_factor, n_bits,
                                         _ptr(seed), _ptr(out)))
    return out.tobytes()


def shard_top_levels(ctx, records, rank):
    """The merge half of the exchange for records gathered by other means: records [world][104]."""
    rec = _u8(records).reshape(-1, RECORD_BYTES)
    world = rec.shape[0]
    bits = world.bit_length() - 1
    o = _top_out(bits)
    _chk(lib().dapol_shard_top_levels(ctx.h, world, rank, _ptr(rec), *[_ptr(x) for x in o]))
    return _top_result(o, bits)


class Tree:
    """dapol_tree: the sparse Merkle sum tree resident in HBM."""

    def __init__(self, ctx, height, leaf_idx, v, r32, pad_seed, enforce_sparsity=False, shard_bits=0, pad_tape=None):
        """height = total tree height; shard_bits > 0 builds only the subtree holding the (global-index) leaves.
        pad_tape (bytes / uint8 array of 64-byte draws; pad_seed is then ignored): dapol_tree_build_tape."""
        self.ctx, self.height, self.shard_bits = ctx, height - shard_bits, shard_bits
        leaf_idx, v = _u64(leaf_idx), _u64(v)
        n = leaf_idx.shape[0]
        r32 = _u8(r32, n, 32)
        self.h = _P()
        if pad_tape is not None:
            tp = _u8(np.frombuffer(bytes(pad_tape), np.uint8)) if len(pad_tape) else np.zeros(64, np.uint8)
            _chk(lib().dapol_tree_build_tape(ctx.h, height, n, _ptr(leaf_idx), _ptr(v), _ptr(r32), _ptr(tp), len(pad_tape) // 64, ctypes.byref(self.h)))
            return
        seed = _u8(np.frombuffer(pad_seed, np.uint8))
        if shard_bits:
            _chk(lib().dapol_tree_build_shard(ctx.h, height, shard_bits, n, _ptr(leaf_idx), _ptr(v), _ptr(r32), _ptr(seed), ctypes.byref(self.h)))
        else:
            _chk(lib().dapol_tree_build(ctx.h, height, n, _ptr(leaf_idx), _ptr(v), _ptr(r32), _ptr(seed), int(enforce_sparsity), ctypes.byref(self.h)))

    def close(self):
        if self.h:
            lib().dapol_tree_destroy(self.h)
            self.h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def root(self):
        C, H, r = np.zeros(32, np.uint8), np.zeros(self.ctx.hb, np.uint8), np.zeros(32, np.uint8)
        v = np.zeros(1, np.uint64)
        _chk(lib().dapol_tree_root(self.h, _ptr(C), _ptr(H), _ptr(v), _ptr(r)))
        return C.tobytes(), H.tobytes(), int(v[0]), r.tobytes()

    def update(self, leaf_idx, v, r32):
        """Dapol::update for a batch of leaves (insert or replace), applied in order."""
        leaf_idx, v, r32 = _u64(leaf_idx), _u64(v), _u8(r32)
        _chk(lib().dapol_tree_update(self.h, leaf_idx.shape[0], _ptr(leaf_idx), _ptr(v), _ptr(r32)))

    def last_update_path(self):
        """0 = the last update rebuilt the tree, 1 = replaced in place, 2 = inserted in place, 3 = both."""
        p = ctypes.c_int32(-1)
        _chk(lib().dapol_tree_last_update_path(self.h, ctypes.byref(p)))
        return int(p.value)

    def node_count(self):
        a, b = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        _chk(lib().dapol_tree_node_count(self.h, _ptr(a), _ptr(b)))
        return int(a[0]), int(b[0])

    def level_nodes(self, level):
        a, b = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        _chk(lib().dapol_tree_level_size(self.h, level, _ptr(a), _ptr(b)))
        n = int(a[0] + b[0])
        idx, v = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        r, C, H = np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8), np.zeros((n, self.ctx.hb), np.uint8)
        pad = np.zeros(n, np.uint8)
        _chk(lib().dapol_tree_level_nodes(self.h, level, _ptr(idx), _ptr(v), _ptr(r), _ptr(C), _ptr(H), _ptr(pad)))
        return idx, v, r, C, H, pad

    def paths(self, leaf_idx):
        leaf_idx = _u64(leaf_idx)
        b, h = leaf_idx.shape[0], self.height
        C, H, r = np.zeros((b, h, 32), np.uint8), np.zeros((b, h, self.ctx.hb), np.uint8), np.zeros((b, h, 32), np.uint8)
        v = np.zeros((b, h), np.uint64)
        _chk(lib().dapol_tree_paths(self.h, b, _ptr(leaf_idx), _ptr(C), _ptr(H), _ptr(v), _ptr(r)))
        return C, H, v, r

    def prove_batch(self, leaf_idx, policy, aggregation_factor, n_bits, nonce_seed):
        """Dapol::generate_proof_batch: one proof for all the leaves.  Returns (sib_level, sib_index, sib_C, sib_H, range blob)."""
        leaf_idx = _u64(leaf_idx)
        k = leaf_idx.shape[0]
        level, index = batch_siblings(self.height, leaf_idx)
        S = level.shape[0]
        es = lib().dapol_entity_proof_size(S, policy, aggregation_factor, n_bits)
        if es == 0:
            raise DapolError(8, "bad policy / aggregation_factor / n_bits")
        C, H = np.zeros((S, 32), np.uint8), np.zeros((S, self.ctx.hb), np.uint8)
        out = np.zeros(es, np.uint8)
        seed = _u8(np.frombuffer(nonce_seed, np.uint8))
        _chk(lib().dapol_prove_batch(self.ctx.h, self.h, k, _ptr(leaf_idx), policy, aggregation_factor, n_bits, _ptr(seed), _ptr(C), _ptr(H),
                                     _ptr(out)))
        return level, index, C, H, out.tobytes()

    def prove_entities(self, leaf_idx, policy, aggregation_factor, n_bits, nonce_seed, upper=None, tape=None):
        """upper = (C[u,32], H[u,32], v[u], r[u,32]) siblings above a shard root, root side first.
        tape (nonce_seed is then ignored): dapol_prove_entities_tape, [b][dapol_entity_tape_slots][64] bytes."""
        leaf_idx = _u64(leaf_idx)
        nu = 0 if upper is None else len(upper[2])
        b, h = leaf_idx.shape[0], self.height + nu
        es = lib().dapol_entity_proof_size(h, policy, aggregation_factor, n_bits)
        C, H = np.zeros((b, h, 32), np.uint8), np.zeros((b, h, self.ctx.hb), np.uint8)
        out = np.zeros((b, max(es, 1)), np.uint8)
        if tape is not None:
            tp = _u8(np.frombuffer(bytes(tape), np.uint8))
            if tp.shape[0] != b * 64 * lib().dapol_entity_tape_slots(h, policy, aggregation_factor, n_bits):
                raise DapolError(8, "the tape must hold dapol_entity_tape_slots draws of 64 bytes per entity")
            _chk(lib().dapol_prove_entities_tape(self.ctx.h, self.h, b, _ptr(leaf_idx), policy, aggregation_factor, n_bits, _ptr(tp), _ptr(C), _ptr(H), _ptr(out)))
            return C, H, out
        seed = _u8(np.frombuffer(nonce_seed, np.uint8))
        if nu:
            uC, uH, uv, ur = _u8(upper[0], nu, 32), _u8(upper[1], nu, 32), _u64(upper[2]), _u8(upper[3], nu, 32)
            _chk(lib().dapol_prove_entities_upper(self.ctx.h, self.h, b, _ptr(leaf_idx), policy, aggregation_factor, n_bits, _ptr(seed), nu,
                                                  _ptr(uC), _ptr(uH), _ptr(uv), _ptr(ur), _ptr(C), _ptr(H), _ptr(out)))
        else:
            _chk(lib().dapol_prove_entities(self.ctx.h, self.h, b, _ptr(leaf_idx), policy, aggregation_factor, n_bits, _ptr(seed), _ptr(C), _ptr(H), _ptr(out)))
        return C, H, out


class Workload:
    """Device-resident bench workload: leaves uploaded once; build() = tree (or shard subtree) build, prove() = one
    padding-policy proof per entity with aggregation_factor = total height."""

    def __init__(self, ctx, height, leaf_idx, v, r32, shard_bits=0):
        self.ctx, self.height, self.shard_bits = ctx, height, shard_bits
        leaf_idx, v = _u64(leaf_idx), _u64(v)
        self.n = leaf_idx.shape[0]
        r32 = _u8(r32, self.n, 32)
        self.h = _P()
        _chk(lib().dapol_workload_create_shard(ctx.h, height, shard_bits, self.n, _ptr(leaf_idx), _ptr(v), _ptr(r32), ctypes.byref(self.h)))

    def close(self):
        if self.h:
            lib().dapol_workload_destroy(self.h)
            self.h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, pad_seed, stats=None):
        st = stats if stats is not None else WorkloadStats()
        ps = _u8(np.frombuffer(pad_seed, np.uint8))
        C, H, r, v = np.zeros(32, np.uint8), np.zeros(32, np.uint8), np.zeros(32, np.uint8), np.zeros(1, np.uint64)
        _chk(lib().dapol_workload_build(self.h, _ptr(ps), _ptr(C), _ptr(H), _ptr(v), _ptr(r), ctypes.byref(st)))
        return (C.tobytes(), H.tobytes(), int(v[0]), r.tobytes()), st

    def prove(self, nonce_seed, n_bits=64, first=0, count=None, upper=None, stats=None, policy=POLICY_PADDING, aggregation_factor=None):
        """One inclusion proof per entity of [first, first + count): padding policy with aggregation_factor = total height by default
        (the headline workload); policy / aggregation_factor select the reference's other shapes (dapol_workload_prove_policy)."""
        st = stats if stats is not None else WorkloadStats()
        count = self.n - first if count is None else count
        ns = _u8(np.frombuffer(nonce_seed, np.uint8))
        agg = self.height if aggregation_factor is None else aggregation_factor
        if upper is None or len(upper[2]) == 0:
            _chk(lib().dapol_workload_prove_policy(self.h, _ptr(ns), n_bits, first, count, policy, agg, 0, None, None, None, None, ctypes.byref(st)))
        else:
            nu = len(upper[2])
            uC, uH, uv, ur = _u8(upper[0], nu, 32), _u8(upper[1], nu, 32), _u64(upper[2]), _u8(upper[3], nu, 32)
            _chk(lib().dapol_workload_prove_policy(self.h, _ptr(ns), n_bits, first, count, policy, agg, nu, _ptr(uC), _ptr(uH), _ptr(uv), _ptr(ur),
                                                   ctypes.byref(st)))
        return st

    def run(self, pad_seed, nonce_seed, n_bits=64, first=0, count=None):
        _, st = self.build(pad_seed)
        return self.prove(nonce_seed, n_bits, first, count, stats=st)

    def tree_handle(self):
        """The (sub)tree of the last build, borrowed (for tree_node_records)."""
        h = _P()
        _chk(lib().dapol_workload_tree(self.h, ctypes.byref(h)))
        return h

    def paths(self, leaf_idx, upper=None, with_nodes=False):
        """Siblings of sampled leaves of the last build: (v, r) and, with_nodes, also the proof nodes (C, H)."""
        leaf_idx = _u64(leaf_idx)
        b = leaf_idx.shape[0]
        nu = 0 if upper is None else len(upper[2])
        sv, sr = np.zeros((b, self.height), np.uint64), np.zeros((b, self.height, 32), np.uint8)
        sC, sH = (np.zeros((b, self.height, 32), np.uint8), np.zeros((b, self.height, 32), np.uint8)) if with_nodes else (None, None)
        uC = _u8(upper[0], nu, 32) if nu else None
        uH = _u8(upper[1], nu, 32) if nu else None
        uv = _u64(upper[2]) if nu else None
        ur = _u8(upper[3], nu, 32) if nu else None
        _chk(lib().dapol_workload_paths(self.h, b, _ptr(leaf_idx), nu, _ptr(uv), _ptr(ur), _ptr(uC), _ptr(uH), _ptr(sv), _ptr(sr), _ptr(sC), _ptr(sH)))
        return (sv, sr, sC, sH) if with_nodes else (sv, sr)

    def proofs(self, first, count, proof_size):
        out = np.zeros((count, proof_size), np.uint8)
        _chk(lib().dapol_workload_proofs(self.h, first, count, _ptr(out)))
        return out
