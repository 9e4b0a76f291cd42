"""ctypes binding of libivfhnsw_hip.so (include/ivfhnsw_hip.h) for the tests and bench.py.

The product is the shared library; this module adds nothing but argument marshalling.  There is no
fallback of any kind: if the library is missing or no gfx950 device is present, calls raise.

The directory name has a hyphen (it mirrors the reference's name), so it is loaded through
`__graft_entry__.load_pkg()` rather than a plain import.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IVFHNSW_HIP_LIB") or os.path.join(_HERE, "libivfhnsw_hip.so")  # override: A/B runs of two builds

OK, ERR_INVALID, ERR_HIP, ERR_STATE, ERR_NOMEM = 0, -1, -2, -3, -4
STAGES = ("opq", "coarse", "lut", "plan", "scan", "select")

# every symbol include/ivfhnsw_hip.h declares
ABI_SYMBOLS = (
    "ivfhnsw_gpu_last_error", "ivfhnsw_gpu_abi_version", "ivfhnsw_gpu_create", "ivfhnsw_gpu_create_view", "ivfhnsw_gpu_destroy",
    "ivfhnsw_gpu_set_stream", "ivfhnsw_gpu_sync", "ivfhnsw_gpu_upload_ivf", "ivfhnsw_gpu_upload_ivf_synthetic",
    "ivfhnsw_gpu_upload_grouping", "ivfhnsw_gpu_upload_quantizer", "ivfhnsw_gpu_search", "ivfhnsw_gpu_search_dev",
    "ivfhnsw_gpu_resolve_keys_dev", "ivfhnsw_gpu_coarse_dev", "ivfhnsw_gpu_coarse", "ivfhnsw_gpu_set_profiling",
    "ivfhnsw_gpu_get_stage_ms", "ivfhnsw_gpu_reset_stage_ms", "ivfhnsw_gpu_last_scan_counts",
    "ivfhnsw_gpu_memory_bytes", "ivfhnsw_gpu_upload_codebooks", "ivfhnsw_gpu_encode",
    "ivfhnsw_gpu_encode_groups", "ivfhnsw_gpu_rotate_dev", "ivfhnsw_gpu_last_scan_kernel",
    "ivfhnsw_gpu_last_stream_dev", "ivfhnsw_gpu_replay_stream_dev", "ivfhnsw_gpu_pq_train", "ivfhnsw_gpu_xty",
    "ivfhnsw_gpu_prepare_latency", "ivfhnsw_gpu_set_batch_split", "ivfhnsw_gpu_last_batch_parts", "ivfhnsw_gpu_search_keys", "ivfhnsw_gpu_resolve_keys", "ivfhnsw_gpu_last_stream",
    "ivfhnsw_gpu_device_count", "ivfhnsw_gpu_knn", "ivfhnsw_gpu_knn_dev", "ivfhnsw_gpu_build_graph", "ivfhnsw_gpu_set_option", "ivfhnsw_gpu_search_sharded",
)


class IvfDesc(C.Structure):
    _fields_ = [("d", C.c_size_t), ("nc", C.c_size_t), ("code_size", C.c_size_t),
                ("offsets", C.c_void_p), ("ids", C.c_void_p), ("codes", C.c_void_p),
                ("norm_codes", C.c_void_p), ("centroid_norms", C.c_void_p), ("pq_centroids", C.c_void_p),
                ("norm_table", C.c_void_p), ("opq_A", C.c_void_p),
                ("shard_rank", C.c_uint32), ("shard_world", C.c_uint32), ("list_owner", C.c_void_p)]


class SearchParams(C.Structure):
    _fields_ = [("nprobe", C.c_size_t), ("max_codes", C.c_size_t), ("efSearch", C.c_size_t),
                ("do_pruning", C.c_int), ("heap_order", C.c_int)]


class IvfHnswError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ivfhnsw_gpu error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load the C-ABI library (once).  Raises if it has not been built: there is no other path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: run __graft_entry__.build() (make -C ivf-hnsw_amd/csrc)" % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64.so.7; one process must not initialise two HIP runtimes
        # (torch then reports "No HIP GPUs are available").  Loading torch first makes this library bind to
        # the runtime torch uses, whichever of the two the caller touches first afterwards.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.ivfhnsw_gpu_last_error.restype = C.c_char_p
        L.ivfhnsw_gpu_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.ivfhnsw_gpu_create_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.ivfhnsw_gpu_destroy.argtypes = [C.c_void_p]
        L.ivfhnsw_gpu_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_sync.argtypes = [C.c_void_p]
        L.ivfhnsw_gpu_upload_ivf.argtypes = [C.c_void_p, C.POINTER(IvfDesc)]
        L.ivfhnsw_gpu_upload_ivf_synthetic.argtypes = [C.c_void_p, C.POINTER(IvfDesc), C.c_uint64]
        L.ivfhnsw_gpu_upload_grouping.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 4
        L.ivfhnsw_gpu_upload_quantizer.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32,
                                                   C.c_void_p, C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_search.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(SearchParams), C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_search_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(SearchParams), C.c_void_p, C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_resolve_keys_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                                   C.c_void_p]
        L.ivfhnsw_gpu_coarse_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                             C.c_void_p]
        L.ivfhnsw_gpu_coarse.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                         C.c_void_p]
        L.ivfhnsw_gpu_upload_codebooks.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_encode_groups.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_rotate_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.ivfhnsw_gpu_get_stage_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.ivfhnsw_gpu_reset_stage_ms.argtypes = [C.c_void_p]
        L.ivfhnsw_gpu_last_scan_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ivfhnsw_gpu_memory_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.ivfhnsw_gpu_last_stream_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                                  C.POINTER(C.c_uint32)]
        L.ivfhnsw_gpu_replay_stream_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32,
                                                    C.c_void_p]
        L.ivfhnsw_gpu_pq_train.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_xty.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_knn.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_int, C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_build_graph.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                              C.c_size_t, C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_knn_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_int, C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        L.ivfhnsw_gpu_search_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.POINTER(SearchParams), C.c_void_p, C.c_void_p]
        L.ivfhnsw_gpu_prepare_latency.argtypes = [C.c_void_p]
        L.ivfhnsw_gpu_set_batch_split.argtypes = [C.c_void_p, C.c_int]
        L.ivfhnsw_gpu_last_batch_parts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ivfhnsw_gpu_last_scan_kernel.argtypes = [C.c_void_p]
        L.ivfhnsw_gpu_last_scan_kernel.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != OK:
        raise IvfHnswError(rc, lib().ivfhnsw_gpu_last_error().decode())


def _np(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _devptr(t):
    """Device pointer of a torch tensor (or a raw int address / None)."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def search_sharded(shards, queries, k, nprobe, max_codes, coarse_ids, coarse_dists, do_pruning=False):
    """ivfhnsw_gpu_search_sharded: the shard step over all shard handles of one process (RCCL merge across devices, host
    merge when they share one); host arrays in and out."""
    q = _np(queries, np.float32)
    q = q.reshape(-1, q.shape[-1])
    nq = q.shape[0]
    cid = _np(coarse_ids, np.uint32).reshape(nq, nprobe)
    cd = _np(coarse_dists, np.float32).reshape(nq, nprobe)
    dist = np.empty((nq, k), np.float32)
    lab = np.empty((nq, k), np.int64)
    arr = (C.c_void_p * len(shards))(*[g._h for g in shards])
    p = SearchParams(nprobe, max_codes, 0, 1 if do_pruning else 0, 0)
    _check(lib().ivfhnsw_gpu_search_sharded(arr, len(shards), nq, k, _ptr(q), _ptr(cid), _ptr(cd), C.byref(p), _ptr(dist),
                                            _ptr(lab)))
    return dist, lab


class GpuIndex:
    """One device-side index (ivfhnsw_gpu handle)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().ivfhnsw_gpu_create(device, C.byref(self._h)))
        self.d = self.nc = self.code_size = 0

    def view(self):
        """A second search context on this index's device tables (own stream and workspace; nothing copied).
        Keep this object alive, and do not upload to it, while the view is in use."""
        v = GpuIndex.__new__(GpuIndex)
        v._h = C.c_void_p()
        _check(lib().ivfhnsw_gpu_create_view(self._h, C.byref(v._h)))
        v.d, v.nc, v.code_size = self.d, self.nc, self.code_size
        v._parent = self
        return v

    def close(self):
        if self._h:
            lib().ivfhnsw_gpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- uploads -------------------------------------------------------------------------------
    def _desc(self, d, code_size, offsets, centroid_norms, pq_centroids, norm_table, opq_A, ids, codes, norm_codes,
              shard_rank, shard_world, list_owner=None):
        keep = {}
        keep["offsets"] = _np(offsets, np.uint64)
        nc = len(keep["offsets"]) - 1
        keep["centroid_norms"] = _np(centroid_norms, np.float32)
        keep["pq_centroids"] = _np(pq_centroids, np.float32)
        keep["norm_table"] = _np(norm_table, np.float32)
        assert keep["centroid_norms"].size == nc
        assert keep["pq_centroids"].size == 256 * d
        assert keep["norm_table"].size == 256
        keep["opq_A"] = None if opq_A is None else _np(opq_A, np.float32)
        keep["ids"] = None if ids is None else _np(ids, np.uint32)
        keep["codes"] = None if codes is None else _np(codes, np.uint8)
        keep["norm_codes"] = None if norm_codes is None else _np(norm_codes, np.uint8)
        keep["list_owner"] = None if list_owner is None else _np(list_owner, np.uint32)
        assert keep["list_owner"] is None or keep["list_owner"].size == nc
        desc = IvfDesc(d, nc, code_size, _ptr(keep["offsets"]), _ptr(keep["ids"]), _ptr(keep["codes"]),
                       _ptr(keep["norm_codes"]), _ptr(keep["centroid_norms"]), _ptr(keep["pq_centroids"]),
                       _ptr(keep["norm_table"]), _ptr(keep["opq_A"]), shard_rank, shard_world,
                       _ptr(keep["list_owner"]))
        self.d, self.nc, self.code_size = d, nc, code_size
        return desc, keep

    def upload_ivf(self, d, code_size, offsets, ids, codes, norm_codes, centroid_norms, pq_centroids, norm_table,
                   opq_A=None, shard_rank=0, shard_world=1, list_owner=None):
        desc, keep = self._desc(d, code_size, offsets, centroid_norms, pq_centroids, norm_table, opq_A, ids, codes,
                                norm_codes, shard_rank, shard_world, list_owner)
        _check(lib().ivfhnsw_gpu_upload_ivf(self._h, C.byref(desc)))

    def upload_ivf_synthetic(self, d, code_size, offsets, centroid_norms, pq_centroids, norm_table, seed, opq_A=None,
                             shard_rank=0, shard_world=1, list_owner=None):
        desc, keep = self._desc(d, code_size, offsets, centroid_norms, pq_centroids, norm_table, opq_A, None, None,
                                None, shard_rank, shard_world, list_owner)
        _check(lib().ivfhnsw_gpu_upload_ivf_synthetic(self._h, C.byref(desc), seed))

    def upload_grouping(self, nsubc, alphas, nn_centroid_idxs, subgroup_sizes, inter_centroid_dists):
        a = _np(alphas, np.float32)
        n = _np(nn_centroid_idxs, np.uint32)
        s = _np(subgroup_sizes, np.uint32)
        i = _np(inter_centroid_dists, np.float32)
        assert a.size == self.nc and n.size == s.size == i.size == self.nc * nsubc
        _check(lib().ivfhnsw_gpu_upload_grouping(self._h, nsubc, _ptr(a), _ptr(n), _ptr(s), _ptr(i)))

    def upload_quantizer(self, link_counts, links, vectors, enterpoint=0):
        c = _np(link_counts, np.uint8)
        v = _np(vectors, np.float32)
        n, d = v.shape
        l = _np(links, np.uint32).reshape(n, -1)
        _check(lib().ivfhnsw_gpu_upload_quantizer(self._h, n, d, l.shape[1], enterpoint, _ptr(c), _ptr(l), _ptr(v)))

    def prepare_latency(self):
        """Build the fat graph of the latency walk (one query per call; see ivfhnsw_gpu_prepare_latency)."""
        _check(lib().ivfhnsw_gpu_prepare_latency(self._h))

    def set_batch_split(self, permille):
        """Batches of >= 8192 queries as two uneven parts on two streams (ivfhnsw_gpu_set_batch_split); 0 = off,
        1..999 = the first part's share, 1000 = chosen per call (the default)."""
        _check(lib().ivfhnsw_gpu_set_batch_split(self._h, int(permille)))

    def last_batch_parts(self):
        """(first, second): queries in the two parts of the last search call; second = 0 when it ran in one part."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(lib().ivfhnsw_gpu_last_batch_parts(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # ---- search --------------------------------------------------------------------------------
    @staticmethod
    def _params(nprobe, max_codes, efSearch, do_pruning, heap_order=False):
        return SearchParams(nprobe, max_codes, efSearch, 1 if do_pruning else 0, 1 if heap_order else 0)

    def search(self, queries, k, nprobe, max_codes, coarse_ids=None, coarse_dists=None, efSearch=0,
               do_pruning=False, heap_order=False):
        """Host arrays in, host arrays out (ivfhnsw_gpu_search)."""
        q = _np(queries, np.float32)
        q = q.reshape(-1, self.d or q.shape[-1])  # before any upload the library itself refuses the call
        nq = q.shape[0]
        cid = None if coarse_ids is None else _np(coarse_ids, np.uint32).reshape(nq, nprobe)
        cd = None if coarse_dists is None else _np(coarse_dists, np.float32).reshape(nq, nprobe)
        dist = np.empty((nq, k), np.float32)
        lab = np.empty((nq, k), np.int64)
        p = self._params(nprobe, max_codes, efSearch, do_pruning, heap_order)
        _check(lib().ivfhnsw_gpu_search(self._h, nq, k, _ptr(q), _ptr(cid), _ptr(cd), C.byref(p), _ptr(dist),
                                        _ptr(lab)))
        return dist, lab

    def search_dev(self, nq, k, d_queries, d_distances, d_labels, nprobe, max_codes, d_coarse_ids=None,
                   d_coarse_dists=None, efSearch=0, do_pruning=False, d_out_keys=None, heap_order=False):
        """Device buffers (torch CUDA tensors), asynchronous on the handle's stream."""
        p = self._params(nprobe, max_codes, efSearch, do_pruning, heap_order)
        _check(lib().ivfhnsw_gpu_search_dev(self._h, nq, k, _devptr(d_queries), _devptr(d_coarse_ids),
                                            _devptr(d_coarse_dists), C.byref(p), _devptr(d_distances),
                                            _devptr(d_labels), _devptr(d_out_keys)))

    def resolve_keys_dev(self, nq, k, d_keys, d_distances, d_labels):
        _check(lib().ivfhnsw_gpu_resolve_keys_dev(self._h, nq, k, _devptr(d_keys), _devptr(d_distances),
                                                  _devptr(d_labels)))

    def last_stream_dev(self, nq, len_cap=0, d_keys=None, d_len=None):
        """Copy out the candidate streams the last search_dev (k > 1, heap_order, out_keys) left: lengths into d_len
        ([nq] int32), the first len_cap keys of each into d_keys ([nq, len_cap] int64).  Returns the stream capacity."""
        cap = C.c_uint32()
        _check(lib().ivfhnsw_gpu_last_stream_dev(self._h, nq, len_cap, _devptr(d_keys), _devptr(d_len), C.byref(cap)))
        return cap.value

    def replay_stream_dev(self, nq, k, d_stream, d_len, cap, d_out_keys):
        _check(lib().ivfhnsw_gpu_replay_stream_dev(self._h, nq, k, _devptr(d_stream), _devptr(d_len), cap,
                                                   _devptr(d_out_keys)))

    def rotate_dev(self, nq, d_queries, d_out):
        """opq_matrix->apply on device buffers (a copy when the index has no OPQ matrix)."""
        _check(lib().ivfhnsw_gpu_rotate_dev(self._h, nq, _devptr(d_queries), _devptr(d_out)))

    def coarse_dev(self, nq, d_queries, nprobe, efSearch, d_coarse_ids, d_coarse_dists):
        _check(lib().ivfhnsw_gpu_coarse_dev(self._h, nq, _devptr(d_queries), nprobe, efSearch,
                                            _devptr(d_coarse_ids), _devptr(d_coarse_dists)))

    def coarse(self, queries, k, efSearch):
        """Host arrays: the HNSW walk alone (k = 1: IndexIVF_HNSW::assign)."""
        q = _np(queries, np.float32)
        q = q.reshape(-1, q.shape[-1])
        ids = np.empty((q.shape[0], k), np.uint32)
        dist = np.empty((q.shape[0], k), np.float32)
        _check(lib().ivfhnsw_gpu_coarse(self._h, q.shape[0], _ptr(q), k, efSearch, _ptr(ids), _ptr(dist)))
        return ids, dist

    # ---- construction side ------------------------------------------------------------------------
    def upload_codebooks(self, d, code_size, pq_centroids, norm_table, opq_A=None):
        pq = _np(pq_centroids, np.float32)
        nt = _np(norm_table, np.float32)
        assert pq.size == 256 * d and nt.size == 256
        A = None if opq_A is None else _np(opq_A, np.float32)
        _check(lib().ivfhnsw_gpu_upload_codebooks(self._h, d, code_size, _ptr(pq), _ptr(nt), _ptr(A)))
        self._enc_M = code_size

    def encode(self, x, precomputed_idx=None, efSearch=0):
        """IndexIVF_HNSW::add_batch up to the append loop (IndexIVF_HNSW.cpp:75-121): (idx, codes, norm_codes)."""
        x = _np(x, np.float32)
        x = x.reshape(-1, x.shape[-1])
        n = x.shape[0]
        pidx = None if precomputed_idx is None else _np(precomputed_idx, np.uint32)
        idx = np.empty(n, np.uint32)
        codes = np.empty((n, self._enc_M), np.uint8)
        ncodes = np.empty(n, np.uint8)
        _check(lib().ivfhnsw_gpu_encode(self._h, n, _ptr(x), _ptr(pidx), efSearch, _ptr(idx), _ptr(codes), _ptr(ncodes)))
        return idx, codes, ncodes

    def encode_groups(self, nsubc, centroid_idx, offsets, x, efSearch, alphas_in=None):
        """IndexIVF_HNSW_Grouping::add_group for many groups (Grouping.cpp:43-125):
        (nn_centroid_idxs [G, nsubc], alphas [G], subcentroid_idxs [n], codes [n, M], norm_codes [n])."""
        cidx = _np(centroid_idx, np.uint32)
        off = _np(offsets, np.uint64)
        G = cidx.shape[0]
        x = _np(x, np.float32)
        x = x.reshape(-1, x.shape[-1]) if x.size else x.reshape(0, 1)
        n = int(off[-1])
        nn = np.empty((G, nsubc), np.uint32)
        alphas = np.zeros(G, np.float32) if alphas_in is None else _np(alphas_in, np.float32).copy()
        sub = np.empty(n, np.uint32)
        codes = np.empty((n, self._enc_M), np.uint8)
        ncodes = np.empty(n, np.uint8)
        _check(lib().ivfhnsw_gpu_encode_groups(self._h, G, nsubc, _ptr(cidx), _ptr(off), _ptr(x), efSearch, _ptr(nn),
                                               _ptr(alphas), _ptr(sub), _ptr(codes), _ptr(ncodes)))
        return nn, alphas, sub, codes, ncodes

    def pq_train(self, x, M, centroids, niter=1):
        """niter Lloyd iterations of ProductQuantizer::train on the device: (centroids [M, 256, d/M], assign [n, M])."""
        x = _np(x, np.float32)
        n, d = x.shape
        c = _np(centroids, np.float32).copy().reshape(M, 256, d // M)
        a = np.empty((n, M), np.uint8)
        _check(lib().ivfhnsw_gpu_pq_train(self._h, n, d, M, _ptr(x), niter, _ptr(c), _ptr(a)))
        return c, a

    def xty(self, X, Y):
        """X^T Y of OPQ's Procrustes step on the matrix cores ([n, d] each -> [d, d])."""
        X = _np(X, np.float32)
        Y = _np(Y, np.float32)
        n, d = X.shape
        out = np.empty((d, d), np.float32)
        _check(lib().ivfhnsw_gpu_xty(self._h, n, d, _ptr(X), _ptr(Y), _ptr(out)))
        return out

    KNN_ALL, KNN_NOT_SELF, KNN_EARLIER = 0, 1, 2

    def knn(self, base, k, queries=None, mode=None):
        """Exact k nearest base rows of every query row (ivfhnsw_gpu_knn; queries=None: of every base row, by default
        itself left out; mode KNN_EARLIER: only rows before it): (ids u32 [nq, k], dists f32 [nq, k]) ascending by
        (dist, id)."""
        if mode is None:
            mode = self.KNN_NOT_SELF if queries is None else self.KNN_ALL
        x = _np(base, np.float32)
        nx, d = x.shape
        q = None if queries is None else _np(queries, np.float32).reshape(-1, d)
        nq = nx if q is None else q.shape[0]
        ids = np.empty((nq, k), np.uint32)
        dist = np.empty((nq, k), np.float32)
        _check(lib().ivfhnsw_gpu_knn(self._h, nq, nx, d, _ptr(q), _ptr(x), k, mode, _ptr(ids), _ptr(dist)))
        return ids, dist

    def knn_dev(self, nq, nx, d, d_queries, d_base, k, d_ids, d_dists=None, exclude_self=False, mode=None):
        """The same on device buffers (torch CUDA tensors), asynchronous on the handle's stream."""
        if mode is None:
            mode = self.KNN_NOT_SELF if exclude_self else self.KNN_ALL
        _check(lib().ivfhnsw_gpu_knn_dev(self._h, nq, nx, d, _devptr(d_queries), _devptr(d_base), k, mode,
                                         _devptr(d_ids), _devptr(d_dists)))

    def build_graph(self, vectors, M=16, maxM=32, ncand=64):
        """hnswlib's addPoint loop for all nodes at once, candidates = the exact ncand nearest earlier nodes
        (ivfhnsw_gpu_build_graph): (counts u8 [n], links u32 [n, maxM])."""
        v = _np(vectors, np.float32)
        n, d = v.shape
        counts = np.zeros(n, np.uint8)
        links = np.zeros((n, maxM), np.uint32)
        _check(lib().ivfhnsw_gpu_build_graph(self._h, n, d, _ptr(v), M, maxM, ncand, _ptr(counts), _ptr(links)))
        return counts, links

    def set_option(self, key, value):
        """Library options (ivfhnsw_gpu_set_option), e.g. ("scan_pipe", 0)."""
        _check(lib().ivfhnsw_gpu_set_option(self._h, key.encode(), int(value)))

    def sync(self):
        _check(lib().ivfhnsw_gpu_sync(self._h))

    def set_stream(self, stream_ptr):
        _check(lib().ivfhnsw_gpu_set_stream(self._h, C.c_void_p(stream_ptr)))

    # ---- measurement ---------------------------------------------------------------------------
    def set_profiling(self, on):
        """True / 1: hipEvents around every stage; 2: only around the scan; False / 0: off."""
        _check(lib().ivfhnsw_gpu_set_profiling(self._h, 2 if on == 2 else (1 if on else 0)))

    def reset_stage_ms(self):
        _check(lib().ivfhnsw_gpu_reset_stage_ms(self._h))

    def stage_ms(self):
        out = {}
        for i, name in enumerate(STAGES):
            ms, n = C.c_double(), C.c_uint64()
            _check(lib().ivfhnsw_gpu_get_stage_ms(self._h, i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def last_scan_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().ivfhnsw_gpu_last_scan_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_scan_kernel(self):
        return lib().ivfhnsw_gpu_last_scan_kernel(self._h).decode()

    def memory_bytes(self):
        a = C.c_uint64()
        _check(lib().ivfhnsw_gpu_memory_bytes(self._h, C.byref(a)))
        return a.value
