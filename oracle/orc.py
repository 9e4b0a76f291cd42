"""ctypes binding of oracle/liborc.so (the CPU restatement of the reference's search path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product.  PARITY UNPINNED (see ivfhnsw_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORC_LIB: another build of the same source (liborc_ofast.so: the float associations of the reference's -Ofast binary,
# tests/test_oracle_float_order.py); the default is the source-order contract
LIB_PATH = os.environ.get("ORC_LIB") or os.path.join(_HERE, "liborc.so")


class OrcIndex(C.Structure):
    _fields_ = [("d", C.c_size_t), ("nc", C.c_size_t), ("code_size", C.c_size_t),
                ("quantizer", C.c_void_p), ("pq_centroids", C.c_void_p), ("norm_table", C.c_float * 256),
                ("opq_A", C.c_void_p), ("do_opq", C.c_int),
                ("nprobe", C.c_size_t), ("max_codes", C.c_size_t), ("efSearch", C.c_size_t),
                ("offsets", C.c_void_p), ("ids", C.c_void_p), ("codes", C.c_void_p), ("norm_codes", C.c_void_p),
                ("centroid_norms", C.c_void_p),
                ("nsubc", C.c_size_t), ("do_pruning", C.c_int), ("alphas", C.c_void_p),
                ("nn_centroid_idxs", C.c_void_p), ("subgroup_sizes", C.c_void_p),
                ("inter_centroid_dists", C.c_void_p)]


class OrcStats(C.Structure):
    _fields_ = [("ncode", C.c_ulonglong), ("nseg", C.c_ulonglong), ("dist_evals", C.c_ulonglong)]


_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, sz = C.c_void_p, C.c_size_t
        L.orc_l2sqr.restype = C.c_float
        L.orc_l2sqr.argtypes = [vp, vp, sz]
        L.orc_inner_prod_table.argtypes = [vp, vp, sz, sz, vp]
        L.orc_opq_apply.argtypes = [vp, vp, sz, vp]
        L.orc_maxheap_heapify.argtypes = [sz, vp, vp]
        L.orc_maxheap_pop.argtypes = [sz, vp, vp]
        L.orc_maxheap_push.argtypes = [sz, vp, vp, C.c_float, C.c_long]
        L.orc_hnsw_new.restype = vp
        L.orc_hnsw_new.argtypes = [sz, sz, sz, sz, sz]
        L.orc_hnsw_free.argtypes = [vp]
        L.orc_hnsw_add_point.argtypes = [vp, vp]
        L.orc_hnsw_from_arrays.restype = vp
        L.orc_hnsw_from_arrays.argtypes = [sz, sz, sz, sz, C.c_uint32, vp, vp, vp]
        L.orc_hnsw_search_knn.restype = sz
        L.orc_hnsw_search_knn.argtypes = [vp, vp, sz, sz, vp, vp]
        for f in ("orc_hnsw_n", "orc_hnsw_d", "orc_hnsw_maxM"):
            getattr(L, f).restype = sz
            getattr(L, f).argtypes = [vp]
        L.orc_hnsw_enterpoint.restype = C.c_uint32
        L.orc_hnsw_enterpoint.argtypes = [vp]
        for f in ("orc_hnsw_counts", "orc_hnsw_links", "orc_hnsw_vectors"):
            getattr(L, f).restype = vp
            getattr(L, f).argtypes = [vp]
        L.orc_hnsw_dist_calc.restype = C.c_ulonglong
        L.orc_hnsw_dist_calc.argtypes = [vp]
        L.orc_hnsw_save.argtypes = [vp, C.c_char_p, C.c_char_p]
        L.orc_hnsw_load.restype = vp
        L.orc_hnsw_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        ip = C.POINTER(OrcIndex)
        sp = C.POINTER(OrcStats)
        L.orc_search_ivf.argtypes = [ip, sz, vp, vp, vp, sp]
        L.orc_search_ivf_coarse.argtypes = [ip, sz, vp, vp, vp, vp, vp, sp]
        L.orc_search_grouping.argtypes = [ip, sz, vp, vp, vp, sp]
        L.orc_search_grouping_coarse.argtypes = [ip, sz, vp, vp, vp, vp, vp, sp]
        L.orc_search_batch.argtypes = [ip, sz, sz, vp, vp, vp, vp, vp, sp, C.c_int]
        L.orc_add_batch_encode.argtypes = [ip, sz, vp, vp, vp, vp, vp, vp]
        L.orc_add_group_encode.argtypes = [ip, sz, C.c_uint32, sz, vp, vp, vp, vp, vp, vp]
        L.orc_add_group_encode.restype = C.c_int
        L.orc_pq_lloyd.argtypes = [sz, sz, sz, vp, sz, vp, vp]
        L.orc_xty.argtypes = [sz, sz, vp, vp, sz, vp]
        L.orc_knn.argtypes = [sz, sz, sz, vp, vp, sz, C.c_int, vp, vp]
        L.orc_hnsw_build_exact.restype = vp
        L.orc_hnsw_build_exact.argtypes = [sz, sz, sz, sz, sz, vp]
        L.orc_compute_centroid_norms.argtypes = [vp, vp]
        L.orc_compute_inter_centroid_dists.argtypes = [vp, sz, vp, vp]
        L.orc_rotate_quantizer.argtypes = [vp, vp]
        L.orc_index_write.argtypes = [ip, C.c_char_p, C.c_int]
        L.orc_index_read.argtypes = [ip, C.c_char_p, C.c_int]
        L.orc_index_free_lists.argtypes = [ip]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def l2sqr(x, y):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    return float(lib().orc_l2sqr(_p(x), _p(y), x.size))


def inner_prod_table(x, pq_centroids, M):
    x = np.ascontiguousarray(x, np.float32)
    c = np.ascontiguousarray(pq_centroids, np.float32)
    out = np.empty((M, 256), np.float32)
    lib().orc_inner_prod_table(_p(x), _p(c), x.size, M, _p(out))
    return out


def opq_apply(A, x):
    A = np.ascontiguousarray(A, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().orc_opq_apply(_p(A), _p(x), x.size, _p(y))
    return y


def pq_lloyd(x, M, centroids, niter=1):
    """niter Lloyd iterations (the oracle's restatement of the code-book training step): (centroids, assign)."""
    x = np.ascontiguousarray(x, np.float32)
    n, d = x.shape
    c = np.ascontiguousarray(centroids, np.float32).copy()
    assert c.size == 256 * d
    assign = np.empty((n, M), np.uint8)
    lib().orc_pq_lloyd(n, d, M, _p(x), niter, _p(c), _p(assign))
    return c, assign


def xty(X, Y, chunk):
    X = np.ascontiguousarray(X, np.float32)
    Y = np.ascontiguousarray(Y, np.float32)
    n, d = X.shape
    C_ = np.empty((d, d), np.float32)
    lib().orc_xty(n, d, _p(X), _p(Y), chunk, _p(C_))
    return C_


def knn(base, k, queries=None, mode=None):
    """The contract of ivfhnsw_gpu_knn: (ids [nq, k], dists [nq, k]) ascending by (dist, id).  mode 0 all rows,
    1 not the row itself (default without queries), 2 only earlier rows."""
    if mode is None:
        mode = 1 if queries is None else 0
    x = np.ascontiguousarray(base, np.float32)
    nx, d = x.shape
    q = None if queries is None else np.ascontiguousarray(queries, np.float32).reshape(-1, d)
    nq = nx if q is None else len(q)
    ids = np.empty((nq, k), np.uint32)
    dist = np.empty((nq, k), np.float32)
    lib().orc_knn(nq, nx, d, _p(q), _p(x), k, mode, _p(ids), _p(dist))
    return ids, dist


class Hnsw:
    """orc_hnsw handle (hnswlib::HierarchicalNSW restatement)."""

    def __init__(self, handle):
        assert handle
        self.h = C.c_void_p(handle)

    @classmethod
    def build(cls, vectors, M=16, efConstruction=500):
        """Serial, reference-identical construction (hnswalg.cpp:212-225), maxM = 2*M
        (IndexIVF_HNSW.cpp:50)."""
        v = np.ascontiguousarray(vectors, np.float32)
        n, d = v.shape
        g = cls(lib().orc_hnsw_new(d, n, M, 2 * M, efConstruction))
        for i in range(n):
            rc = lib().orc_hnsw_add_point(g.h, _p(v[i]))
            assert rc == 0
        return g

    @classmethod
    def build_exact(cls, vectors, M=16, maxM=32, ncand=64):
        """The serial insertion loop with exact candidates (the contract of ivfhnsw_gpu_build_graph)."""
        v = np.ascontiguousarray(vectors, np.float32)
        n, d = v.shape
        return cls(lib().orc_hnsw_build_exact(d, n, M, maxM, ncand, _p(v)))

    @classmethod
    def from_arrays(cls, counts, links, vectors, M, enterpoint=0):
        v = np.ascontiguousarray(vectors, np.float32)
        n, d = v.shape
        c = np.ascontiguousarray(counts, np.uint8)
        l = np.ascontiguousarray(links, np.uint32).reshape(n, -1)
        return cls(lib().orc_hnsw_from_arrays(d, n, M, l.shape[1], enterpoint, _p(c), _p(l), _p(v)))

    @classmethod
    def load(cls, path_info, path_data, path_edges):
        h = lib().orc_hnsw_load(path_info.encode(), path_data.encode(), path_edges.encode())
        if not h:
            raise IOError("orc_hnsw_load failed")
        return cls(h)

    def save(self, path_info, path_edges):
        assert lib().orc_hnsw_save(self.h, path_info.encode(), path_edges.encode()) == 0

    @property
    def n(self):
        return lib().orc_hnsw_n(self.h)

    @property
    def d(self):
        return lib().orc_hnsw_d(self.h)

    @property
    def maxM(self):
        return lib().orc_hnsw_maxM(self.h)

    @property
    def enterpoint(self):
        return lib().orc_hnsw_enterpoint(self.h)

    @property
    def counts(self):
        return _view(lib().orc_hnsw_counts(self.h), (self.n,), np.uint8)

    @property
    def links(self):
        return _view(lib().orc_hnsw_links(self.h), (self.n, self.maxM), np.uint32)

    @property
    def vectors(self):
        return _view(lib().orc_hnsw_vectors(self.h), (self.n, self.d), np.float32)

    def search_knn(self, query, ef, k):
        q = np.ascontiguousarray(query, np.float32)
        ids = np.empty(k, np.uint32)
        dist = np.empty(k, np.float32)
        r = lib().orc_hnsw_search_knn(self.h, _p(q), ef, k, _p(ids), _p(dist))
        return ids[:r].copy(), dist[:r].copy()

    def rotate(self, A):
        A = np.ascontiguousarray(A, np.float32)
        lib().orc_rotate_quantizer(self.h, _p(A))

    def centroid_norms(self):
        out = np.empty(self.n, np.float32)
        lib().orc_compute_centroid_norms(self.h, _p(out))
        return out

    def inter_centroid_dists(self, nn_idx):
        nn = np.ascontiguousarray(nn_idx, np.uint32)
        out = np.empty(nn.shape, np.float32)
        lib().orc_compute_inter_centroid_dists(self.h, nn.shape[1], _p(nn), _p(out))
        return out

    def free(self):
        if self.h:
            lib().orc_hnsw_free(self.h)
            self.h = C.c_void_p()


class Index:
    """orc_index over numpy arrays (kept alive by this object)."""

    def __init__(self, d, code_size, quantizer, pq_centroids, norm_table, offsets, ids, codes, norm_codes,
                 centroid_norms, opq_A=None, nsubc=0, alphas=None, nn_centroid_idxs=None, subgroup_sizes=None,
                 inter_centroid_dists=None):
        self.keep = dict(
            pq=np.ascontiguousarray(pq_centroids, np.float32),
            offsets=np.ascontiguousarray(offsets, np.uint64),
            ids=np.ascontiguousarray(ids, np.uint32),
            codes=np.ascontiguousarray(codes, np.uint8),
            norm_codes=np.ascontiguousarray(norm_codes, np.uint8),
            cn=np.ascontiguousarray(centroid_norms, np.float32),
            A=None if opq_A is None else np.ascontiguousarray(opq_A, np.float32),
            alphas=None if alphas is None else np.ascontiguousarray(alphas, np.float32),
            nn=None if nn_centroid_idxs is None else np.ascontiguousarray(nn_centroid_idxs, np.uint32),
            sg=None if subgroup_sizes is None else np.ascontiguousarray(subgroup_sizes, np.uint32),
            icd=None if inter_centroid_dists is None else np.ascontiguousarray(inter_centroid_dists, np.float32),
        )
        self.quantizer = quantizer
        k = self.keep
        ix = OrcIndex()
        ix.d, ix.nc, ix.code_size = d, len(k["offsets"]) - 1, code_size
        ix.quantizer = quantizer.h if quantizer is not None else None
        ix.pq_centroids = _p(k["pq"])
        nt = np.ascontiguousarray(norm_table, np.float32)
        assert nt.size == 256
        for i in range(256):
            ix.norm_table[i] = float(nt[i])
        ix.opq_A = _p(k["A"])
        ix.do_opq = 0 if opq_A is None else 1
        ix.offsets, ix.ids, ix.codes, ix.norm_codes = _p(k["offsets"]), _p(k["ids"]), _p(k["codes"]), _p(k["norm_codes"])
        ix.centroid_norms = _p(k["cn"])
        ix.nsubc = nsubc
        ix.alphas, ix.nn_centroid_idxs = _p(k["alphas"]), _p(k["nn"])
        ix.subgroup_sizes, ix.inter_centroid_dists = _p(k["sg"]), _p(k["icd"])
        self.ix = ix
        self.d, self.nsubc = d, nsubc

    def set_params(self, nprobe, max_codes, efSearch, do_pruning=False):
        self.ix.nprobe, self.ix.max_codes, self.ix.efSearch = nprobe, max_codes, efSearch
        self.ix.do_pruning = 1 if do_pruning else 0

    def add_batch_encode(self, x, precomputed_idx=None):
        """IndexIVF_HNSW.cpp:75-121 for the rows of x: (idx, codes, norm_codes, norms).  Uses efSearch of
        set_params for the assignment when precomputed_idx is None."""
        x = np.ascontiguousarray(x, np.float32)
        n = x.shape[0]
        pidx = None if precomputed_idx is None else np.ascontiguousarray(precomputed_idx, np.uint32)
        idx = np.empty(n, np.uint32)
        codes = np.empty((n, self.ix.code_size), np.uint8)
        ncodes = np.empty(n, np.uint8)
        norms = np.empty(n, np.float32)
        lib().orc_add_batch_encode(C.byref(self.ix), n, _p(x), _p(pidx), _p(idx), _p(codes), _p(ncodes), _p(norms))
        return idx, codes, ncodes, norms

    def add_group_encode(self, nsubc, centroid_idx, data):
        """IndexIVF_HNSW_Grouping.cpp:43-125 for one group: (nn_centroid_idxs, alpha, subcentroid_idxs, codes,
        norm_codes).  efSearch of set_params drives searchKnn(centroid, nsubc + 1)."""
        data = np.ascontiguousarray(data, np.float32).reshape(-1, self.d)
        n = data.shape[0]
        nn = np.empty(nsubc, np.uint32)
        alpha = np.zeros(1, np.float32)
        sub = np.empty(n, np.uint32)
        codes = np.empty((n, self.ix.code_size), np.uint8)
        ncodes = np.empty(n, np.uint8)
        rc = lib().orc_add_group_encode(C.byref(self.ix), nsubc, int(centroid_idx), n, _p(data), _p(nn), _p(alpha),
                                        _p(sub), _p(codes), _p(ncodes))
        assert rc == 0, "the walk found fewer than nsubc + 1 centroids"
        return nn, alpha[0], sub, codes, ncodes

    def search(self, x, k=1):
        """One query through the reference's single-query entry point."""
        x = np.ascontiguousarray(x, np.float32)
        dist = np.empty(k, np.float32)
        lab = np.empty(k, np.int64)
        st = OrcStats()
        f = lib().orc_search_grouping if self.nsubc else lib().orc_search_ivf
        f(C.byref(self.ix), k, _p(x), _p(dist), _p(lab), C.byref(st))
        return dist, lab, st

    def search_coarse(self, x, cidx, cdist, k=1):
        x = np.ascontiguousarray(x, np.float32)
        cidx = np.ascontiguousarray(cidx, np.uint32)
        cdist = np.ascontiguousarray(cdist, np.float32)
        assert cidx.size == self.ix.nprobe
        dist = np.empty(k, np.float32)
        lab = np.empty(k, np.int64)
        st = OrcStats()
        f = lib().orc_search_grouping_coarse if self.nsubc else lib().orc_search_ivf_coarse
        f(C.byref(self.ix), k, _p(x), _p(cidx), _p(cdist), _p(dist), _p(lab), C.byref(st))
        return dist, lab, st

    def search_batch(self, x, k=1, nthreads=1, want_coarse=True):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, self.d)
        nq = x.shape[0]
        dist = np.empty((nq, k), np.float32)
        lab = np.empty((nq, k), np.int64)
        npb = self.ix.nprobe
        cid = np.empty((nq, npb), np.uint32) if want_coarse else None
        cd = np.empty((nq, npb), np.float32) if want_coarse else None
        st = OrcStats()
        lib().orc_search_batch(C.byref(self.ix), nq, k, _p(x), _p(dist), _p(lab), _p(cid), _p(cd), C.byref(st),
                               nthreads)
        return dist, lab, cid, cd, st

    def write(self, path):
        assert lib().orc_index_write(C.byref(self.ix), path.encode(), 1 if self.nsubc else 0) == 0


def read_index(path, grouping):
    """Read a .index file into numpy arrays (copies; the C allocations are released)."""
    ix = OrcIndex()
    if lib().orc_index_read(C.byref(ix), path.encode(), 1 if grouping else 0) != 0:
        raise IOError("orc_index_read failed: %s" % path)
    nc, N, cs, ns = ix.nc, 0, ix.code_size, ix.nsubc
    off = _view(ix.offsets, (nc + 1,), np.uint64).copy()
    N = int(off[-1])
    out = dict(d=ix.d, nc=nc, code_size=cs, nsubc=ns, offsets=off,
               ids=_view(ix.ids, (N,), np.uint32).copy() if N else np.zeros(0, np.uint32),
               codes=_view(ix.codes, (N, cs), np.uint8).copy() if N else np.zeros((0, cs), np.uint8),
               norm_codes=_view(ix.norm_codes, (N,), np.uint8).copy() if N else np.zeros(0, np.uint8),
               centroid_norms=_view(ix.centroid_norms, (nc,), np.float32).copy())
    if grouping:
        out.update(alphas=_view(ix.alphas, (nc,), np.float32).copy(),
                   nn_centroid_idxs=_view(ix.nn_centroid_idxs, (nc, ns), np.uint32).copy(),
                   subgroup_sizes=_view(ix.subgroup_sizes, (nc, ns), np.uint32).copy(),
                   inter_centroid_dists=_view(ix.inter_centroid_dists, (nc, ns), np.float32).copy())
    lib().orc_index_free_lists(C.byref(ix))
    return out
