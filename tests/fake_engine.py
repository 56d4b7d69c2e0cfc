"""CPU stand-in for galah_amd.distributed.HipEngine, backed by the oracle.  TEST CODE ONLY: it
lets the world_size>1 exchange logic (shard ranges, all-gathers, tile dealing, candidate gather)
run under gloo without a GPU.  Same method names and array layouts as HipEngine."""
import types

import numpy as np
import torch

import oracle
from galah_amd._lib import PAIR_DTYPE
from galah_amd.distributed import tile_pairs_of_rank

PT = 8  # tile edge of pair_intersect_tile at s <= 1087 (ghip_pair_geometry)


class FakeEngine:
    def __init__(self, kmer=21, sketch_size=200, ani_k=15, ani_c=125, ani_chunk=20000):
        self.kmer, self.s = kmer, sketch_size
        self.ani_k, self.ani_c, self.ani_chunk = ani_k, ani_c, ani_chunk
        self.streams = []

    def load_synthetic(self, seed, members, first, count, length, sub_rate):
        self.streams = [oracle.synth_genome(seed, (first + i) // members, (first + i) % members, length, sub_rate)
                        for i in range(count)]

    @property
    def local_bases(self):
        return sum(len(s) for s in self.streams)

    def sketch_local(self, block):
        hashes = np.full((block, self.s), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
        lens = np.zeros(block, dtype=np.uint32)
        for i, st in enumerate(self.streams):
            sk = oracle.sketch_bytes(st, self.kmer, self.s, 0)
            hashes[i, : len(sk)] = sk
            lens[i] = len(sk)
        return torch.from_numpy(hashes.view(np.int64)), torch.from_numpy(lens.view(np.int32))

    def sketches_to_host(self, hashes, lens):
        return hashes.numpy().view(np.uint64), lens.numpy().view(np.uint32)

    def precluster(self, hashes, lens, n, min_ani, rank, world):
        """-> (pairs, replicated) like HipEngine.precluster; `replicate` mimics the join form (whole list everywhere)."""
        h, l = self.sketches_to_host(hashes, lens)
        rows, compared = [], 0
        replicated = bool(getattr(self, "replicate", False)) and world > 1
        share_rank, share_world = (0, 1) if replicated else (rank, world)
        for ti, tj in tile_pairs_of_rank(n, PT, share_rank, share_world):
            for i in range(ti * PT, min((ti + 1) * PT, n)):
                for j in range(tj * PT, min((tj + 1) * PT, n)):
                    if i >= j:
                        continue
                    compared += 1
                    c, t = oracle.raw_distance(h[i, : l[i]], h[j, : l[j]])
                    ani = oracle.mash_ani(c, t, self.kmer)
                    if ani >= float(np.float32(min_ani)):
                        rows.append((i, j, c, t, np.float32(ani)))
        if replicated:
            compared = compared // world + (1 if rank < compared % world else 0)
        self.last_pairs_compared = compared
        out = np.array(rows, dtype=PAIR_DTYPE) if rows else np.zeros(0, dtype=PAIR_DTYPE)
        return np.sort(out, order=["i", "j"]), replicated

    def ani_build_local(self):
        caps, cnts, glens, sh, sc, ct = [], [], [], [], [], []
        for st in self.streams:
            a = oracle.AniSketch.from_bytes(st, self.ani_k, self.ani_c, self.ani_chunk)
            h, ch = a.seeds(), a.chunks()
            cap = len(st) // self.ani_c + len(st) // (10 * self.ani_c) + 256
            pad = cap - len(h)
            sh.append(np.concatenate([h, np.zeros(pad, np.uint64)]))
            sc.append(np.concatenate([ch.astype(np.uint16), np.zeros(pad, np.uint16)]))
            nch = (len(st) + self.ani_chunk - 1) // self.ani_chunk
            ct.append(np.bincount(ch, minlength=nch).astype(np.uint32))
            caps.append(cap); cnts.append(len(h)); glens.append(len(st))
        meta = {"glen": np.array(glens, np.uint64), "cap": np.array(caps, np.uint64),
                "cnt": np.array(cnts, np.uint32)}
        arrs = {"seed_hash": np.concatenate(sh) if sh else np.zeros(0, np.uint64),
                "seed_chunk": np.concatenate(sc) if sc else np.zeros(0, np.uint16),
                "bin_start": np.zeros(len(caps) * 16385, np.uint32),
                "chunk_total": np.concatenate(ct) if ct else np.zeros(0, np.uint32)}
        lay = types.SimpleNamespace(n_seed_slots=len(arrs["seed_hash"]), n_bin_slots=len(arrs["bin_start"]),
                                    n_chunk_slots=len(arrs["chunk_total"]))
        idx = {"meta": meta, "arrs": arrs}
        return idx, meta, lay

    def ani_export(self, idx, lay):
        a = idx["arrs"]
        return {"seed_code": torch.from_numpy(a["seed_hash"].view(np.int64)),  # the stand-in keeps the 64-bit hash
                "seed_chunk": torch.from_numpy(a["seed_chunk"].view(np.int16)),
                "bin_start": torch.from_numpy(a["bin_start"].view(np.int32)),
                "chunk_total": torch.from_numpy(a["chunk_total"].view(np.int32))}

    def ani_wrap(self, meta, arrs):
        return {"meta": meta, "arrs": {"seed_hash": arrs["seed_code"].numpy().view(np.uint64),
                                        "seed_chunk": arrs["seed_chunk"].numpy().view(np.uint16),
                                        "bin_start": arrs["bin_start"].numpy().view(np.uint32),
                                        "chunk_total": arrs["chunk_total"].numpy().view(np.uint32)}}

    def _genome(self, idx, g):
        m, a = idx["meta"], idx["arrs"]
        s0 = int(np.sum(m["cap"][:g]))
        nch = [(int(x) + self.ani_chunk - 1) // self.ani_chunk for x in m["glen"]]
        c0 = int(sum(nch[:g]))
        cnt = int(m["cnt"][g])
        return (a["seed_hash"][s0: s0 + cnt], a["seed_chunk"][s0: s0 + cnt], int(m["glen"][g]),
                a["chunk_total"][c0: c0 + nch[g]])

    def _direction(self, q, r):
        qh, qc, ql, qt = q
        hit = np.isin(qh, r[0])
        T = np.bincount(qc, minlength=len(qt)).astype(np.uint64)
        assert np.array_equal(T, qt.astype(np.uint64)), "chunk totals lost in the exchange"
        M = np.bincount(qc, weights=hit, minlength=len(qt)).astype(np.uint64)
        al = (T >= 1) & (M * 10000 >= 510 * T)
        bases = sum(min((c + 1) * self.ani_chunk, ql) - c * self.ani_chunk for c in np.nonzero(al)[0])
        return [(int(m), int(t)) for m, t in zip(M[al], T[al])], int(bases)

    def ani_pairs(self, idx, pairs, min_af):
        from fractions import Fraction
        out = np.zeros(len(pairs), np.float32)
        for x, p in enumerate(pairs):
            q, r = self._genome(idx, int(p["i"])), self._genome(idx, int(p["j"]))
            f1, b1 = self._direction(q, r)
            f2, b2 = self._direction(r, q)
            fr = sorted(f1 + f2, key=lambda mt: Fraction(mt[0], mt[1]))
            afq, afr = b1 / q[2], b2 / r[2]
            if fr and not (afq < min_af and afr < min_af):
                m, t = fr[(len(fr) - 1) // 2]
                out[x] = np.float32(float("%.2f" % (100.0 * (m / t) ** (1.0 / self.ani_k))))
        return out

    def cluster(self, n, pairs, pair_ani, ani_threshold):
        from galah_amd.engine import cluster_pairs
        return cluster_pairs(n, pairs, ani_threshold, pair_ani, False)
