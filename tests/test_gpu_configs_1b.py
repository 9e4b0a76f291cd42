"""BASELINE.json configs[2], [3] and [4] at FULL size on one MI355X (1B codes generated on the device, the reference's
993 127 / 999 973 centroids), each checked two ways:

  * a 1024-query sample against the oracle -- labels, distance bits and the scored-code count `ncode`.  The oracle
    needs the lists as host arrays; a full host copy of 1B codes is 21 GB and most of a minute, so only the lists the
    oracle's OWN coarse walk probes for the sample are materialised (synth.synthetic_codes_sparse: the rest of the
    address range stays untouched zero pages);
  * size-independent properties on a larger batch (idempotence, batch-order independence, max_codes monotonicity,
    top-1 = head of top-k, pruning scans a subset, shards partition the scan).

Parameters: examples/run_sift1b.sh:37-43 (the two paper points (32, 10000, 80) and (64, 30000, 100)),
examples/run_sift1b_grouping_OPQ.sh:7-53 (nsubc 64, OPQ, pruning, (32, 10000, 80)),
examples/run_deep1b_grouping_OPQ.sh:38-47 / run_deep1b_OPQ.sh (96-d, 999 973 centroids, OPQ, (128, 100000, 130)).
"""
import numpy as np
import pytest

import synth
from oracle import orc

pytestmark = pytest.mark.gpu

N1B = 1_000_000_000
NSAMPLE = 1024
NTHREADS = 16


def _tables(seed, nc, d, kind):
    tb = synth.make_throughput_tables(seed, nc, d, 16, N1B, kind=kind)
    # the coarse graph as the bench builds it: hnswlib's insertion loop with exact candidates, on the device
    import __graft_entry__ as ge
    gb = ge.load_pkg().GpuIndex(0)
    counts, links = gb.build_graph(tb["centroids"], 16, 32, 64)
    gb.close()
    cn = (tb["centroids"].astype(np.float64) ** 2).sum(1).astype(np.float32)
    tb.update(counts=counts, links=links, centroid_norms=cn, code_seed=seed + 7)
    return tb


def _queries(tb, n, seed, noise):
    rng = np.random.default_rng(seed)
    return (tb["centroids"][rng.choice(tb["nc"], n)] + rng.normal(0, noise, size=(n, tb["d"]))).astype(np.float32)


def _oracle_sample(tb, vectors, q, nprobe, max_codes, ef, opq_A=None, gt=None, pruning=False):
    """The oracle on the sample queries over a sparse host view of the device's corpus."""
    graph = orc.Hnsw.from_arrays(tb["counts"], tb["links"], vectors, 16, 0)
    arrays = synth.synthetic_codes_sparse(tb["code_seed"], tb["offsets"], tb["code_size"])
    kw = {}
    if gt is not None:
        kw = dict(nsubc=gt["nsubc"], alphas=gt["alphas"], nn_centroid_idxs=gt["nn_centroid_idxs"],
                  subgroup_sizes=gt["subgroup_sizes"], inter_centroid_dists=gt["inter_centroid_dists"])
    ox = orc.Index(tb["d"], tb["code_size"], graph, tb["pq_centroids"], tb["norm_table"], tb["offsets"], arrays[0],
                   arrays[1], arrays[2], tb["centroid_norms"], opq_A=opq_A, **kw)
    ox.set_params(nprobe, max_codes, ef, do_pruning=pruning)
    # pass 1 over the still empty lists: only to learn which lists the oracle's own walk probes
    _, _, cid, _, _ = ox.search_batch(q, 1, NTHREADS)
    probed = cid.ravel()
    synth.synthetic_codes_sparse(tb["code_seed"], tb["offsets"], tb["code_size"], probed[probed < tb["nc"]], into=arrays)
    ref_d, ref_l, cid, cd, st = ox.search_batch(q, 1, NTHREADS)
    graph.free()
    return ref_d, ref_l, cid, cd, st


def _check_sample(g, tb, vectors, q, nprobe, max_codes, ef, **kw):
    pruning = kw.get("pruning", False)
    ref_d, ref_l, cid, cd, st = _oracle_sample(tb, vectors, q, nprobe, max_codes, ef, **kw)
    dist, lab = g.search(q, 1, nprobe, max_codes, efSearch=ef, do_pruning=pruning)
    assert g.last_scan_counts()[0] == st.ncode
    assert np.array_equal(lab, ref_l)
    assert np.array_equal(dist.view(np.uint32), ref_d.view(np.uint32))
    # the reference's search2 split (coarse stage supplied by the caller) must agree as well
    d2, l2 = g.search(q[:256], 1, nprobe, max_codes, coarse_ids=cid[:256], coarse_dists=cd[:256], do_pruning=pruning)
    assert np.array_equal(l2, ref_l[:256]) and np.array_equal(d2.view(np.uint32), ref_d[:256].view(np.uint32))
    return ref_d, ref_l, cid, cd


def _check_properties(g, q, nprobe, max_codes, ef, n_total, pruning=False):
    d1, l1 = g.search(q, 1, nprobe, max_codes, efSearch=ef, do_pruning=pruning)
    n1 = g.last_scan_counts()[0]
    d2, l2 = g.search(q, 1, nprobe, max_codes, efSearch=ef, do_pruning=pruning)
    assert np.array_equal(l1, l2) and np.array_equal(d1.view(np.uint32), d2.view(np.uint32))
    assert (l1 >= 0).all() and (l1 < n_total).all()
    perm = np.random.default_rng(0).permutation(len(q))
    d3, l3 = g.search(q[perm], 1, nprobe, max_codes, efSearch=ef, do_pruning=pruning)
    assert np.array_equal(l3, l1[perm]) and np.array_equal(d3.view(np.uint32), d1[perm].view(np.uint32))
    for i in range(16):  # one query per call (split scan) == batched scan
        ds, ls = g.search(q[i], 1, nprobe, max_codes, efSearch=ef, do_pruning=pruning)
        assert ls[0, 0] == l1[i, 0] and ds[0, 0] == d1[i, 0]
    # max_codes is a prefix rule: a larger bound only appends (sub)lists to the scan
    dm, _ = g.search(q, 1, nprobe, 3 * max_codes, efSearch=ef, do_pruning=pruning)
    assert g.last_scan_counts()[0] >= n1
    if not pruning:  # with pruning the threshold itself moves with the pass-1 horizon (Grouping.cpp:256-261)
        assert (dm <= d1).all()
    # top-1 is the head of top-10; results ascending; ten distinct labels
    d10, l10 = g.search(q[:512], 10, nprobe, max_codes, efSearch=ef, do_pruning=pruning)
    assert np.array_equal(l10[:, :1], l1[:512]) and np.array_equal(d10[:, :1].view(np.uint32), d1[:512].view(np.uint32))
    assert (np.diff(d10, axis=1) >= 0).all()
    assert all(len(set(row.tolist())) == 10 for row in l10)
    return d1, l1, n1


@pytest.fixture(scope="module")
def sift1b(pkg):
    """993 127-node graph + 1B PQ16 codes on the device: the SIFT1B shape of configs[2] and of the metric line."""
    tb = _tables(1234, 993127, 128, "sift")
    g = pkg.GpuIndex(0)
    g.upload_ivf_synthetic(128, 16, tb["offsets"], tb["centroid_norms"], tb["pq_centroids"], tb["norm_table"],
                           tb["code_seed"])
    g.upload_quantizer(tb["counts"], tb["links"], tb["centroids"], 0)
    yield g, tb
    g.close()


@pytest.mark.parametrize("nprobe,max_codes,ef", [(64, 30000, 100), (32, 10000, 80)])
def test_config2_sift1b_ivfadc(sift1b, nprobe, max_codes, ef):
    """configs[2] (IndexIVF_HNSW::search, nprobe 64, max_codes 30000, efSearch 100) and the metric's own point."""
    g, tb = sift1b
    q = _queries(tb, 4096, 21, 12.0)
    _check_sample(g, tb, tb["centroids"], q[:NSAMPLE], nprobe, max_codes, ef)
    _check_properties(g, q, nprobe, max_codes, ef, N1B)
    # with the bound off the scan is exactly the nprobe lists the walk returns
    g.search(q[:512], 1, nprobe, 10 ** 9, efSearch=ef)
    sizes = np.diff(tb["offsets"].astype(np.int64))
    ids, dist = g.coarse(q[:512], nprobe, ef)
    assert g.last_scan_counts()[0] == int(sizes[ids.astype(np.int64)].sum())
    assert (np.diff(dist, axis=1) >= 0).all()


@pytest.fixture(scope="module")
def grouping1b(pkg, sift1b):
    """configs[3] on ONE GPU: the same lists as sub-groups of 64 neighbour centroids, OPQ, pruning tables."""
    _, tb = sift1b
    gt = synth.make_grouping_tables(tb["seed"] + 3, tb, 64)
    A = synth.random_rotation(np.random.default_rng(tb["seed"] + 4), 128)
    vectors = synth.rotated_vectors(tb["centroids"], A)  # rotate_quantizer (IndexIVF_HNSW.cpp:789-800)
    g = pkg.GpuIndex(0)
    g.upload_ivf_synthetic(128, 16, tb["offsets"], tb["centroid_norms"], tb["pq_centroids"], tb["norm_table"],
                           tb["code_seed"], opq_A=A)
    g.upload_quantizer(tb["counts"], tb["links"], vectors, 0)
    g.upload_grouping(64, gt["alphas"], gt["nn_centroid_idxs"], gt["subgroup_sizes"], gt["inter_centroid_dists"])
    yield g, tb, gt, A, vectors
    g.close()


def test_config3_grouping_pruning_opq(grouping1b):
    g, tb, gt, A, vectors = grouping1b
    q = _queries(tb, 4096, 22, 12.0)
    _check_sample(g, tb, vectors, q[:NSAMPLE], 32, 10000, 80, opq_A=A, gt=gt, pruning=True)
    _, _, n_pruned = _check_properties(g, q, 32, 10000, 80, N1B, pruning=True)
    # without pruning every sub-group of the visited lists is scored (Grouping.cpp:308): the IVFADC code count
    _check_sample(g, tb, vectors, q[:256], 32, 10000, 80, opq_A=A, gt=gt, pruning=False)


def _as_eight_shards(pkg, g, tb, vectors, A, gt, d, nprobe, max_codes, ef, pruning, owner, q):
    """The index of handle `g` again as 8 shard handles on the one GPU: every shard holds the lists the owner table gives
    it, all derive the same global plan, the packed keys are MIN-merged (torch.minimum here, an RCCL all-reduce on the
    node: ivf-hnsw_amd/distributed.py) and the owner resolves each label.  Must equal the one-handle result bit for bit."""
    import torch
    world, nq = 8, len(q)
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(q).to(dev)
    d_ref = torch.empty((nq, 1), dtype=torch.float32, device=dev)
    l_ref = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    g.search_dev(nq, 1, d_q, d_ref, l_ref, nprobe, max_codes, efSearch=ef, do_pruning=pruning)
    g.sync()
    n_ref = g.last_scan_counts()[0]
    # coarse stage once, on rotated queries, handed to every shard (the search2 split)
    xr = torch.empty_like(d_q)
    g.rotate_dev(nq, d_q, xr)
    d_cid = torch.empty((nq, nprobe), dtype=torch.int32, device=dev)
    d_cd = torch.empty((nq, nprobe), dtype=torch.float32, device=dev)
    g.coarse_dev(nq, xr, nprobe, ef, d_cid, d_cd)
    g.sync()
    merged, shards, total, per_rank = None, [], 0, []
    for r in range(world):
        s = pkg.GpuIndex(0)
        s.upload_ivf_synthetic(d, 16, tb["offsets"], tb["centroid_norms"], tb["pq_centroids"], tb["norm_table"],
                               tb["code_seed"], opq_A=A, shard_rank=r, shard_world=world, list_owner=owner)
        s.upload_quantizer(tb["counts"], tb["links"], vectors, 0)
        if gt is not None:
            s.upload_grouping(64, gt["alphas"], gt["nn_centroid_idxs"], gt["subgroup_sizes"], gt["inter_centroid_dists"])
        dd = torch.empty((nq, 1), dtype=torch.float32, device=dev)
        ll = torch.empty((nq, 1), dtype=torch.int64, device=dev)
        kk = torch.empty((nq, 1), dtype=torch.int64, device=dev)
        s.search_dev(nq, 1, d_q, dd, ll, nprobe, max_codes, d_coarse_ids=d_cid, d_coarse_dists=d_cd, do_pruning=pruning,
                     d_out_keys=kk)
        s.sync()
        per_rank.append(s.last_scan_counts()[0])
        total += per_rank[-1]
        merged = kk if merged is None else torch.minimum(merged, kk)
        shards.append((s, dd, ll))
    assert total == n_ref  # the shards partition the scanned codes exactly
    assert min(per_rank) > 0
    label = torch.full((nq, 1), -1, dtype=torch.int64, device=dev)
    for s, dd, ll in shards:
        s.resolve_keys_dev(nq, 1, merged, dd, ll)
        s.sync()
        label = torch.maximum(label, ll)
    assert torch.equal(label, l_ref)
    assert torch.equal(shards[-1][1].view(torch.int32), d_ref.view(torch.int32))
    for s, _, _ in shards:
        s.close()


def test_config3_as_eight_shards(pkg, grouping1b):
    """configs[3] "codes sharded across 8 MI355X", rehearsed as 8 shard handles behind the load-balanced SPATIAL owner
    table the multi-GPU bench can run with (distributed.partition_lists over the 993 127 centroids)."""
    import importlib
    D = importlib.import_module("ivfhnsw_amd.distributed")
    g, tb, gt, A, vectors = grouping1b
    sizes = np.diff(tb["offsets"].astype(np.int64))
    load = D.expected_list_load(sizes, tb["counts"], tb["links"])
    owner = D.partition_lists(tb["centroids"], sizes, 8, "spatial", load=load)
    assert set(owner.tolist()) == set(range(8))
    _as_eight_shards(pkg, g, tb, vectors, A, gt, 128, 32, 10000, 80, True, owner, _queries(tb, 2048, 23, 12.0))


def test_config4_deep1b_opq(pkg):
    """configs[4] on ONE GPU: 1B x 96-d, 999 973 centroids, OPQ, PQ16, (nprobe, max_codes, efSearch) =
    (128, 100000, 130).  For a throughput corpus the synthetic centroids simply ARE the rotated ones."""
    tb = _tables(4321, 999973, 96, "deep")
    A = synth.random_rotation(np.random.default_rng(tb["seed"] + 4), 96)
    g = pkg.GpuIndex(0)
    try:
        g.upload_ivf_synthetic(96, 16, tb["offsets"], tb["centroid_norms"], tb["pq_centroids"], tb["norm_table"],
                               tb["code_seed"], opq_A=A)
        g.upload_quantizer(tb["counts"], tb["links"], tb["centroids"], 0)
        q = _queries(tb, 2048, 24, 0.03)
        _check_sample(g, tb, tb["centroids"], q[:NSAMPLE], 128, 100000, 130, opq_A=A)
        _check_properties(g, q, 128, 100000, 130, N1B)
        # configs[4] "8 MI355X shard": the same index as 8 shard handles, lists dealt by a hash of the list number
        owner = (((np.arange(tb["nc"], dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(9)) % np.uint64(8))
        _as_eight_shards(pkg, g, tb, tb["centroids"], A, None, 96, 128, 100000, 130, False, owner.astype(np.uint32),
                         q[:1024])
    finally:
        g.close()
