"""A CPU stand-in for one GpuIndex shard, so that the REAL ivf-hnsw_amd/distributed.py::ShardedSearcher can run over gloo
ranks without a GPU (tests/test_distributed_cpu.py).  Test infrastructure: the shard's plan + ADC scan are the oracle's
arithmetic in numpy (IndexIVF_HNSW.cpp:267-292 with GLOBAL scan positions, lists of other ranks skipped but counted),
the coarse stage is the oracle's walk.  It implements exactly the methods ShardedSearcher calls, on CPU torch tensors.
"""
import numpy as np

from oracle import orc

SIGN = np.uint64(0x8000000000000000)
F = np.float32


class FakeShard:
    def __init__(self, c, ox, rank, owner, pack_keys, unpack_keys, stream_cap=8192):
        self.c, self.ox, self.rank, self.owner = c, ox, rank, np.asarray(owner)
        self.pack_keys, self.unpack_keys, self.cap = pack_keys, unpack_keys, stream_cap
        self.streams, self.labels = None, None
        self.calls = []

    # -- plumbing ShardedSearcher expects ------------------------------------------------------------------------
    def set_stream(self, ptr):
        self.calls.append(("set_stream", ptr))

    def sync(self):
        pass

    def rotate_dev(self, n, d_q, d_out):
        d_out[:n].copy_(d_q[:n])   # no OPQ in the CPU corpus

    def coarse_dev(self, n, d_x, nprobe, ef, d_cid, d_cd):
        import torch
        self.ox.set_params(nprobe, 1, ef)    # only the coarse stage of the oracle is used
        _, _, cid, cd, _ = self.ox.search_batch(np.ascontiguousarray(d_x[:n].numpy()), k=1)
        d_cid[:n].copy_(torch.from_numpy(cid.astype(np.int32)))
        d_cd[:n].copy_(torch.from_numpy(cd))

    # -- the shard step -----------------------------------------------------------------------------------------
    def _scan(self, q, cid, cd, max_codes):
        """Every code of the owned lists in scan order: unsigned keys per query + position -> label."""
        c = self.c
        off = c["offsets"].astype(np.int64)
        M = c["code_size"]
        streams, labels = [], []
        for i in range(len(q)):
            tab = orc.inner_prod_table(q[i], c["pq_centroids"], M)
            ncode, ks, lab = 0, [], {}
            for p, cc in enumerate(cid[i]):
                cc = int(cc)
                n = off[cc + 1] - off[cc]
                if n == 0:
                    continue
                if self.owner[cc] == self.rank:
                    codes = c["codes"][off[cc]:off[cc + 1]]
                    s = np.zeros(n, F)
                    for m in range(M):
                        s = (s + tab[m, codes[:, m]]).astype(F)
                    term1 = F(cd[i, p] - c["centroid_norms"][cc])
                    dist = ((term1 + c["norm_table"][c["norm_codes"][off[cc]:off[cc + 1]]]).astype(F) - F(2) * s).astype(F)
                    vpos = (ncode + np.arange(n)).astype(np.uint32)
                    ks.append(self.pack_keys(dist, vpos).view(np.uint64) ^ SIGN)
                    for j in range(n):
                        lab[int(vpos[j])] = int(c["ids"][off[cc] + j])
                ncode += n
                if ncode >= max_codes:
                    break
            streams.append(np.concatenate(ks) if ks else np.zeros(0, np.uint64))
            labels.append(lab)
        return streams, labels

    def search_dev(self, nq, k, d_q, d_dist, d_lab, nprobe, max_codes, d_coarse_ids=None, d_coarse_dists=None,
                   do_pruning=False, d_out_keys=None, heap_order=False, efSearch=0):
        import torch
        assert d_coarse_ids is not None and d_out_keys is not None and not do_pruning
        cid = d_coarse_ids[:nq].numpy().astype(np.uint32)
        cd = d_coarse_dists[:nq].numpy()
        self.streams, self.labels = self._scan(d_q[:nq].numpy(), cid, cd, max_codes)
        init = np.uint64((0x7f7fffff | 0x80000000) << 32)
        loc = np.full((nq, k), init, np.uint64)
        for i, st in enumerate(self.streams):
            srt = np.sort(st)[:k]
            loc[i, :len(srt)] = srt
        d_out_keys.copy_(torch.from_numpy((loc ^ SIGN).view(np.int64).copy()))

    def resolve_keys_dev(self, nq, k, d_keys, d_dist, d_lab):
        import torch
        keys = d_keys.numpy().reshape(nq, k)
        dd, vv = self.unpack_keys(keys.reshape(-1))
        dd, vv = dd.reshape(nq, k), vv.reshape(nq, k)
        lab = np.full((nq, k), -1, np.int64)
        init = np.uint64((0x7f7fffff | 0x80000000) << 32)
        ukeys = keys.view(np.uint64) ^ SIGN
        for i in range(nq):
            for j in range(k):
                if ukeys[i, j] < init:
                    lab[i, j] = self.labels[i].get(int(vv[i, j]), -1)
        d_dist.copy_(torch.from_numpy(dd.astype(np.float32)))
        d_lab.copy_(torch.from_numpy(lab))

    def last_stream_dev(self, nq, len_cap=0, d_keys=None, d_len=None):
        import torch
        if d_len is not None:
            d_len.copy_(torch.tensor([len(s) for s in self.streams], dtype=torch.int32))
        if d_keys is not None:
            pad = np.zeros((nq, len_cap), np.uint64)
            for i, s in enumerate(self.streams):
                n = min(len(s), len_cap)
                pad[i, :n] = s[:n]
            d_keys.copy_(torch.from_numpy(pad.view(np.int64).copy()))
        return self.cap

    def replay_stream_dev(self, nq, k, d_stream, d_len, cap, d_out_keys):
        """faiss's heap replayed over the merged stream (IndexIVF_HNSW.cpp:265,285-288); out = the heap ARRAY as keys."""
        import torch
        ms = d_stream.numpy().view(np.uint64)
        out = np.empty((nq, k), np.int64)
        init_key = self.pack_keys(np.array([np.finfo(F).max], F), np.array([0], np.uint32))[0]
        for i in range(nq):
            hv = np.empty(k, np.float32)
            hl = np.empty(k, np.int64)      # heap "ids" = scan positions here
            orc.lib().orc_maxheap_heapify(k, orc._p(hv), orc._p(hl))
            seq = ms[i, :int(d_len[i])]
            dd, vv = self.unpack_keys((seq ^ SIGN).view(np.int64))
            for dj, vj in zip(dd, vv):
                if dj < hv[0]:
                    orc.lib().orc_maxheap_pop(k, orc._p(hv), orc._p(hl))
                    orc.lib().orc_maxheap_push(k, orc._p(hv), orc._p(hl), float(dj), int(vj))
            for j in range(k):
                out[i, j] = init_key if hl[j] < 0 else self.pack_keys(hv[j:j + 1], np.array([hl[j]], np.uint32))[0]
        d_out_keys.copy_(torch.from_numpy(out))
