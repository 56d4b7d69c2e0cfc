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

    # ---- persisted sketch matrix (no counterpart in the reference's finch back-end -- skani's sketched
    # reference set, src/skani.rs:502-565, and docs/preludes/cluster_prelude.md:13-15 are the workflow it serves)
    def distances_and_save(self, genome_fasta_paths: Sequence[str], matrix_path: str) -> SortedPairGenomeDistanceCache:
        """distances(), and the sketch matrix with the genome names persisted for later incremental runs."""
        if self.low_memory:
            raise RuntimeError("Low-memory clustering currently only supported with skani preclusterer")
        ctx = self._context()
        try:
            sk = ctx.sketch_files(list(genome_fasta_paths), self.kmer_length, self.num_kmers, 0, self.io_threads)
        except Exception as e:  # finch.rs:72
            raise RuntimeError(f"Failed to sketch genomes with finch: {e}") from e
        sk.save(matrix_path, list(genome_fasta_paths), 0)
        pairs = ctx.precluster(sk, np.float32(self.min_ani))
        sk.free()
        self.last_pairs = pairs
        cache = SortedPairGenomeDistanceCache.from_pairs(pairs)
        cache._pairs = pairs
        return cache

    def distances_incremental(self, matrix_path: str, new_genome_fasta_paths: Sequence[str], saved_pairs: Optional[np.ndarray] = None,
                              save_to: Optional[str] = None):
        """Incremental dereplication: the genome list is [the saved matrix's genomes ..., the new files ...]; only the NEW files
        are read and sketched, and the pair stage runs on the (new x all) rectangle.  Returns (genome names, cache): with
        `saved_pairs` (the earlier run's edge list, last_pairs) the cache is the one distances() over all the files would
        produce; without, it holds the pairs that touch a new genome.  save_to: persist the grown matrix."""
        if self.low_memory:
            raise RuntimeError("Low-memory clustering currently only supported with skani preclusterer")
        ctx = self._context()
        saved, names, seed = ctx.sketches_load_named(matrix_path)
        if (saved.kmer, saved.size, seed) != (self.kmer_length, self.num_kmers, 0):
            raise RuntimeError(f"sketch matrix {matrix_path} was made with k={saved.kmer} s={saved.size} seed={seed}, "
                               f"not k={self.kmer_length} s={self.num_kmers} seed=0")
        try:
            fresh = ctx.sketch_files(list(new_genome_fasta_paths), self.kmer_length, self.num_kmers, 0, self.io_threads)
        except Exception as e:  # finch.rs:72
            raise RuntimeError(f"Failed to sketch genomes with finch: {e}") from e
        n_old = len(saved)
        both = ctx.sketches_concat(saved, fresh)
        saved.free(); fresh.free()
        all_names = names + list(new_genome_fasta_paths)
        if save_to is not None:
            both.save(save_to, all_names, 0)
        pairs = ctx.precluster_from(both, n_old, np.float32(self.min_ani))
        both.free()
        if saved_pairs is not None and len(saved_pairs):
            assert int(saved_pairs["j"].max()) < n_old, "saved_pairs refer to genomes beyond the saved matrix"
            pairs = np.concatenate([np.ascontiguousarray(saved_pairs, dtype=pairs.dtype), pairs])
            pairs = pairs[np.lexsort((pairs["j"], pairs["i"]))]
        self.last_pairs = pairs
        cache = SortedPairGenomeDistanceCache.from_pairs(pairs)
        cache._pairs = pairs
        return all_names, cache

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
