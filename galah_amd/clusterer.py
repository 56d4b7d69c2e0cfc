"""clusterer::cluster mirror (reference src/clusterer.rs:14-152).

Same argument list and the same panics (raised as RuntimeError); the O(N^2) host loops run in
the C++ host clusterer behind ghip_cluster, and a back-end that offers `calculate_ani_indices` is
asked for the clusterer's ANI lazily, in batched rounds (ghip_cluster_lazy: only the precluster
pairs that touch a representative; short lists in one round).
Returns Vec<Vec<usize>> with the representative first in each inner list.
"""
from __future__ import annotations

import contextlib
import gc
import logging
from typing import List, Optional, Sequence

import numpy as np

from ._lib import PAIR_DTYPE
from .engine import cluster_pairs, cluster_pairs_lazy

log = logging.getLogger("galah_amd")


@contextlib.contextmanager
def _cycle_collector_paused():
    """cluster() builds thousands of small lists and no reference cycles; a full cycle collection that happens to be
    due in the middle of it walks every long-lived object of the process (20-35 ms with torch imported) -- more
    than the GPU work of a thousand genomes.  Hold the collector off for the duration of the call."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _pairs_of(cache) -> np.ndarray:
    if getattr(cache, "_dict", 0) is None and getattr(cache, "_src", None) is not None:
        return cache._src   # still the back-end's edge list, nothing was inserted or looked up
    items = cache.items()
    out = np.zeros(len(items), dtype=PAIR_DTYPE)
    for x, ((i, j), v) in enumerate(items):
        if v is None:
            raise RuntimeError("precluster cache holds None: not produced by a precluster back-end")
        out[x] = (i, j, 0, 0, v)
    return out


def _one_ingest(preclusterer, clusterer) -> bool:
    from .ani import HipAniClusterer
    from .finch import FinchPreclusterer
    if not (isinstance(preclusterer, FinchPreclusterer) and isinstance(clusterer, HipAniClusterer)):
        return False
    if preclusterer.low_memory:
        return False  # distances() raises the reference's refusal
    return clusterer._ctx is None or clusterer._ctx is preclusterer._context()


def _distances_and_index(genomes, preclusterer, clusterer):
    from .ani import ANI_CHUNK, ANI_K
    from .cache import SortedPairGenomeDistanceCache
    ctx = preclusterer._context()
    clusterer._ctx = ctx
    try:  # batched inside the library: inputs larger than HBM work
        sk, idx, _stats = ctx.sketch_and_index_files(list(genomes), preclusterer.kmer_length, preclusterer.num_kmers, 0, ANI_K,
                                                     clusterer.seed_compression, ANI_CHUNK,
                                                     max(preclusterer.io_threads, clusterer.io_threads))
    except Exception as e:  # finch.rs:72
        raise RuntimeError(f"Failed to sketch genomes with finch: {e}") from e
    pairs = ctx.precluster(sk, np.float32(preclusterer.min_ani))
    sk.free()
    clusterer.adopt_index(idx, None, genomes)
    cache = SortedPairGenomeDistanceCache.from_pairs(pairs)
    preclusterer.last_pairs = pairs
    return cache


def cluster(genomes: Sequence[str], preclusterer, clusterer, cluster_contigs: bool = False,
            contig_names: Optional[Sequence[str]] = None,
            reference_genomes: Optional[Sequence[str]] = None) -> List[List[int]]:
    with _cycle_collector_paused():
        return _cluster(genomes, preclusterer, clusterer, cluster_contigs, contig_names, reference_genomes)


def _cluster(genomes, preclusterer, clusterer, cluster_contigs, contig_names, reference_genomes) -> List[List[int]]:
    clusterer.initialise()
    preclusterer_name = preclusterer.method_name()
    clusterer_name = clusterer.method_name()
    log.info("Preclustering with %s and clustering with %s", preclusterer_name, clusterer_name)

    skip_clusterer = False
    if clusterer_name == preclusterer_name:  # clusterer.rs:32-36
        log.info("Preclustering and clustering methods are the same, so reusing ANI values")
        skip_clusterer = True
    if cluster_contigs:  # clusterer.rs:38-44
        if preclusterer_name == "finch":
            raise RuntimeError(f"{preclusterer_name} does not support contig comparisons.")
        skip_clusterer = True

    if reference_genomes is not None:  # clusterer.rs:47-54
        cache = preclusterer.distances_with_references(genomes, reference_genomes)
    elif cluster_contigs:
        cache = preclusterer.distances_contigs(genomes, contig_names)
    elif not skip_clusterer and _one_ingest(preclusterer, clusterer):
        # both back-ends are HIP: read every FASTA once and take the MinHash sketches AND the ANI index from one
        # pass over the bases (the reference reads each genome for finch, then twice per ANI pair for skani)
        cache = _distances_and_index(genomes, preclusterer, clusterer)
    else:
        cache = preclusterer.distances(genomes)

    n = len(contig_names) if cluster_contigs else len(genomes)
    pairs = _pairs_of(cache)
    threshold = np.float32(clusterer.get_ani_threshold())
    if skip_clusterer:
        return cluster_pairs(n, pairs, threshold, None, True)
    if hasattr(clusterer, "calculate_ani_indices"):
        # edge indices are positions in `genomes`: the clusterer's device index serves them only if it was built for
        # this list in this order (not for another list of the same length, not grown by calculate_ani())
        if not clusterer.prepared_for(genomes):
            clusterer.prepare(list(genomes))
        # the ANI of a precluster pair is only ever needed when it touches a representative (clusterer.rs:194-204,
        # 377-405): asked lazily, one batch per round of the greedy clusterer
        if hasattr(clusterer, "cluster_on_index"):   # the HIP clusterer: the rounds stay in native code (ghip_cluster_index)
            clusters, asked = clusterer.cluster_on_index(n, pairs, threshold)
        else:
            idx = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32) if len(pairs) else np.zeros((0, 2), np.uint32)
            clusters, asked = cluster_pairs_lazy(n, pairs, threshold, lambda edges: clusterer.calculate_ani_indices(idx[edges]))
        clusterer.last_pairs_asked = asked
        return clusters
    return cluster_pairs(n, pairs, threshold, None, False,
                         ani_callback=lambda a, b: clusterer.calculate_ani(genomes[a], genomes[b]))


class GalahClusterer:
    """The library entry CoverM and `galah process` use (src/cluster_argument_parsing.rs:108-115, 1514-1530): a carrier of
    the genome list and the two back-ends whose cluster() is clusterer::cluster -- Vec<Vec<usize>>, element 0 of every inner
    list the representative (src/cluster_argument_parsing.rs:730)."""

    def __init__(self, genome_fasta_paths: Sequence[str], preclusterer, clusterer, cluster_contigs: bool = False,
                 contig_names: Optional[Sequence[str]] = None, reference_genomes: Optional[Sequence[str]] = None):
        self.genome_fasta_paths = list(genome_fasta_paths)
        self.preclusterer = preclusterer
        self.clusterer = clusterer
        self.cluster_contigs = cluster_contigs
        self.contig_names = contig_names
        self.reference_genomes = list(reference_genomes) if reference_genomes is not None else None

    def cluster(self) -> List[List[int]]:
        return cluster(self.genome_fasta_paths, self.preclusterer, self.clusterer, self.cluster_contigs, self.contig_names,
                       self.reference_genomes)
