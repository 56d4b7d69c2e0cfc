"""FinchPreclusterer mirror (reference src/finch.rs:4-97) on the MI355X back-end.

Same constructor fields, same method names, same refusals; `distances` runs
ghip_sketch_files + ghip_precluster and fills a SortedPairGenomeDistanceCache with the exact
f32 values the reference would store.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from .cache import SortedPairGenomeDistanceCache
from .engine import Context


class FinchPreclusterer:
    def __init__(self, min_ani: float, num_kmers: int = 1000, kmer_length: int = 21, low_memory: bool = False,
                 ctx: Optional[Context] = None, io_threads: int = 1):
        self.min_ani = np.float32(min_ani)  # fraction, not percentage (finch.rs:5-6)
        self.num_kmers = num_kmers
        self.kmer_length = kmer_length
        self.low_memory = low_memory
        self._ctx = ctx
        self.io_threads = io_threads
        self.last_pairs: Optional[np.ndarray] = None

    def _context(self) -> Context:
        if self._ctx is None:
            self._ctx = Context(0)
        return self._ctx

    def distances(self, genome_fasta_paths: Sequence[str]) -> SortedPairGenomeDistanceCache:
        if self.low_memory:  # finch.rs:14-15
            raise RuntimeError("Low-memory clustering currently only supported with skani preclusterer")
        return distances(genome_fasta_paths, self.min_ani, self.num_kmers, self.kmer_length,
                         ctx=self._context(), io_threads=self.io_threads, _keep=self)

    def distances_contigs(self, _genome_fasta_paths, _contig_names) -> SortedPairGenomeDistanceCache:
        return SortedPairGenomeDistanceCache()  # finch.rs:26-33

    def distances_with_references(self, _genome_fasta_paths, _reference_genomes):
        raise RuntimeError("Reference genome clustering currently only supported with skani preclusterer")  # finch.rs:40

    def method_name(self) -> str:
        return "finch"


def distances(genome_fasta_paths: Sequence[str], min_ani: float, num_kmers: int, kmer_length: int,
              ctx: Optional[Context] = None, io_threads: int = 1, _keep=None) -> SortedPairGenomeDistanceCache:
    """finch::distances(genome_fasta_paths, min_ani, num_kmers, kmer_length) (finch.rs:48-97)."""
    ctx = ctx or Context(0)
    try:
        sk = ctx.sketch_files(list(genome_fasta_paths), kmer_length, num_kmers, 0, io_threads)
    except Exception as e:  # finch.rs:72
        raise RuntimeError(f"Failed to sketch genomes with finch: {e}") from e
    pairs = ctx.precluster(sk, np.float32(min_ani))
    sk.free()
    cache = SortedPairGenomeDistanceCache.from_pairs(pairs)
    cache._pairs = pairs  # sorted (i, j) edge list for the host clusterer
    if _keep is not None:
        _keep.last_pairs = pairs
    return cache
