"""ClusterDistanceFinder mirror for the batched GPU ANI stage.

Mirrors SkaniClusterer (reference src/skani.rs:689-716): initialise() asserts a percent
threshold, calculate_ani(fasta1, fasta2) -> Some(percent f32) with 0.0 below the aligned-fraction
gate.  Instead of one `skani dist` subprocess per pair (src/skani.rs:718-788) the genomes are
sketched once into an HBM-resident FracMinHash index and pairs are answered in batches.
The estimator is build-defined; skani parity is unpinned (DESIGN.md "ANI").
"""
from __future__ import annotations

import threading
from typing import Dict, Optional, Sequence

import numpy as np

from .engine import AniIndex, Context

ANI_K = 15
ANI_CHUNK = 20000


class HipAniClusterer:
    def __init__(self, threshold: float, min_aligned_threshold: float = 0.15, small_genomes: bool = False,
                 ctx: Optional[Context] = None, io_threads: int = 1, seed_compression: Optional[int] = None):
        self.threshold = np.float32(threshold)                       # percent (skani.rs:203-209)
        self.min_aligned_threshold = np.float32(min_aligned_threshold)  # fraction (skani.rs:734: * 100 for the CLI)
        self.small_genomes = small_genomes
        self._seed_compression = seed_compression  # override of the FracMinHash density (1 = every 15-mer is a seed)
        self._ctx = ctx
        self.io_threads = io_threads
        self._index: Optional[AniIndex] = None
        self._genomes = None
        self._path_index: Dict[str, int] = {}
        self._lock = threading.Lock()  # calculate_ani may be called from many threads (clusterer.rs:14 `C: Sync`)

    @property
    def seed_compression(self) -> int:
        if self._seed_compression is not None:
            return int(self._seed_compression)
        return 30 if self.small_genomes else 125  # skani --small-genomes ~ -c 30

    def _context(self) -> Context:
        if self._ctx is None:
            self._ctx = Context(0)
        return self._ctx

    def initialise(self) -> None:
        assert self.threshold > 1.0  # skani.rs:696-698

    def method_name(self) -> str:
        return "hipani"

    def get_ani_threshold(self) -> np.float32:
        return self.threshold

    # ---- batched interface used by galah_amd.clusterer.cluster
    def prepare(self, genome_fasta_paths: Sequence[str]) -> None:
        ctx = self._context()
        if self._index is not None:
            self._index.free()
        self._genomes = ctx.genomes_from_files(list(genome_fasta_paths), self.io_threads)
        self._index = ctx.ani_index_build(self._genomes, ANI_K, self.seed_compression, ANI_CHUNK)
        self._path_index = {p: i for i, p in enumerate(genome_fasta_paths)}

    def prepared_for(self, genome_fasta_paths: Sequence[str]) -> bool:
        """True when genome i of the device index is genome_fasta_paths[i] for every i."""
        pi = self._path_index
        return (self._index is not None and len(pi) == len(genome_fasta_paths)
                and all(pi.get(g) == i for i, g in enumerate(genome_fasta_paths)))

    def prepare_from_genomes(self, genomes, names: Optional[Sequence[str]] = None) -> None:
        ctx = self._context()
        self._genomes = genomes
        self._index = ctx.ani_index_build(genomes, ANI_K, self.seed_compression, ANI_CHUNK)
        self._path_index = {p: i for i, p in enumerate(names)} if names is not None else {}

    def adopt_index(self, index: AniIndex, genomes, genome_fasta_paths: Sequence[str]) -> None:
        """Take an index built elsewhere (the fused sketch + index pass of galah_amd.clusterer.cluster)."""
        if self._index is not None:
            self._index.free()
        self._genomes, self._index = genomes, index
        self._path_index = {p: i for i, p in enumerate(genome_fasta_paths)}

    def calculate_ani_indices(self, pairs: np.ndarray) -> np.ndarray:
        assert self._index is not None, "call prepare() first"
        return self._context().ani_pairs(self._index, pairs, float(self.min_aligned_threshold))

    def cluster_on_index(self, n_genomes: int, pairs: np.ndarray, threshold) -> tuple:
        """clusterer::cluster over the prepared index, whole in native code (ghip_cluster_index) -> (clusters, pairs asked)."""
        assert self._index is not None, "call prepare() first"
        clusters, st = self._context().cluster_index(self._index, n_genomes, pairs, threshold, float(self.min_aligned_threshold))
        return clusters.tolist(), st["asked"]   # (clusterer::cluster's Vec<Vec<usize>>: plain lists for the caller)

    # ---- the trait method
    def calculate_ani(self, fasta1: str, fasta2: str) -> Optional[np.float32]:
        # The reference calls this from rayon workers (clusterer.rs:267-270,283-293,375-399).  One lock for the whole
        # call: re-indexing frees the index other callers would be reading, and the context runs one call at a time anyway.
        with self._lock:
            if fasta1 not in self._path_index or fasta2 not in self._path_index:
                self.prepare(list(dict.fromkeys(list(self._path_index) + [fasta1, fasta2])))
            pair = np.array([[self._path_index[fasta1], self._path_index[fasta2]]], dtype=np.uint32)
            return np.float32(self.calculate_ani_indices(pair)[0])  # always Some(..) (skani.rs:709)
