"""clusterer::cluster mirror (reference src/clusterer.rs:14-152).

Same argument list and the same panics (raised as RuntimeError); the O(N^2) host loops run in
the C++ host clusterer behind ghip_cluster, and the clusterer's ANI is requested in ONE batch
for all precluster pairs when the back-end offers `calculate_ani_indices`.
Returns Vec<Vec<usize>> with the representative first in each inner list.
"""
from __future__ import annotations

import logging
from typing import List, Optional, Sequence

import numpy as np

from ._lib import PAIR_DTYPE
from .engine import cluster_pairs

log = logging.getLogger("galah_amd")


def _pairs_of(cache) -> np.ndarray:
    pairs = getattr(cache, "_pairs", None)
    if pairs is not None and len(pairs) == len(cache):
        return pairs
    items = cache.items()
    out = np.zeros(len(items), dtype=PAIR_DTYPE)
    for x, ((i, j), v) in enumerate(items):
        if v is None:
            raise RuntimeError("precluster cache holds None: not produced by a precluster back-end")
        out[x] = (i, j, 0, 0, v)
    return out


def cluster(genomes: Sequence[str], preclusterer, clusterer, cluster_contigs: bool = False,
            contig_names: Optional[Sequence[str]] = None,
            reference_genomes: Optional[Sequence[str]] = None) -> List[List[int]]:
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
    else:
        cache = preclusterer.distances(genomes)

    n = len(contig_names) if cluster_contigs else len(genomes)
    pairs = _pairs_of(cache)
    threshold = np.float32(clusterer.get_ani_threshold())
    if skip_clusterer:
        return cluster_pairs(n, pairs, threshold, None, True)
    if hasattr(clusterer, "calculate_ani_indices"):
        if getattr(clusterer, "_index", None) is None or len(getattr(clusterer, "_path_index", {})) != len(genomes):
            clusterer.prepare(list(genomes))
        idx = np.stack([pairs["i"], pairs["j"]], axis=1).astype(np.uint32) if len(pairs) else np.zeros((0, 2), np.uint32)
        pair_ani = clusterer.calculate_ani_indices(idx) if len(pairs) else np.zeros(0, np.float32)
        return cluster_pairs(n, pairs, threshold, pair_ani, False)
    return cluster_pairs(n, pairs, threshold, None, False,
                         ani_callback=lambda a, b: clusterer.calculate_ani(genomes[a], genomes[b]))
