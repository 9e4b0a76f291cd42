"""Writers of the reference's on-disk inputs, used to hand synthetic corpora to the C++ host library
and to the reference's own drivers: .fvecs/.bvecs/.ivecs (utils.h:85-127), faiss ProductQuantizer and
LinearTransform files (layouts: include/faiss/index_io.h)."""
import struct

import numpy as np


def write_xvecs(path, arr):
    """uint32 dim + dim elements per record; dtype of `arr` decides .fvecs (f32) / .bvecs (u8) / .ivecs (i32/u32)."""
    arr = np.ascontiguousarray(arr)
    n, d = arr.shape
    rec = np.empty((n, 4 + d * arr.itemsize), np.uint8)
    rec[:, :4] = np.frombuffer(struct.pack("<I", d), np.uint8)
    rec[:, 4:] = arr.view(np.uint8).reshape(n, d * arr.itemsize)
    rec.tofile(path)


def write_pq(path, d, M, centroids):
    c = np.ascontiguousarray(centroids, np.float32).reshape(-1)
    assert c.size == d * 256
    with open(path, "wb") as f:
        f.write(struct.pack("<QQQQ", d, M, 8, c.size))
        f.write(c.tobytes())


def write_opq(path, A):
    A = np.ascontiguousarray(A, np.float32)
    d_out, d_in = A.shape
    with open(path, "wb") as f:
        f.write(b"LTra")
        f.write(struct.pack("<?", False))
        f.write(struct.pack("<Q", A.size))
        f.write(A.tobytes())
        f.write(struct.pack("<Q", 0))
        f.write(struct.pack("<ii?", d_in, d_out, True))


def dump_corpus(c, dirpath, queries=None):
    """Everything a driver loads for corpus dict `c` (tests/synth.py).  Returns the paths.
    NOTE: c["graph"] holds ROTATED centroids when OPQ is on; the centroid file must hold the originals, which the
    driver rotates itself (rotate_quantizer), so they are passed separately."""
    import os
    from oracle import orc
    p = {k: os.path.join(dirpath, v) for k, v in dict(
        centroids="centroids.fvecs", info="hnsw.info", edges="hnsw.edges", pq="pq.dat", norm_pq="norm_pq.dat",
        opq="opq.dat", index="corpus.index", queries="queries.fvecs").items()}
    write_xvecs(p["centroids"], c["centroids"])
    c["graph"].save(p["info"], p["edges"])
    write_pq(p["pq"], c["d"], c["code_size"], c["pq_centroids"])
    write_pq(p["norm_pq"], 1, 1, c["norm_table"])
    if c["opq_A"] is not None:
        write_opq(p["opq"], c["opq_A"])
    else:
        p["opq"] = "-"
    synth_index(c).write(p["index"])
    write_xvecs(p["queries"], np.ascontiguousarray(c["queries"] if queries is None else queries, np.float32))
    return p


def synth_index(c):
    import synth
    return synth.oracle_index(c)
