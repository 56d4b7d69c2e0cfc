"""Genome ordering by quality (host glue the clusterer's result depends on).

`clusterer::cluster` picks representatives greedily in genome-list order, and galah sorts that
list by a quality score before clustering (reference src/cluster_argument_parsing.rs:1070-1100,
formula "Parks2020_reduced", the default per src/lib.rs:82):
    completeness*100 - 5*contamination*100 - 5*num_contigs/100 - 5*num_ambiguous_bases/100000
with completeness / contamination as f32 fractions widened to f64, sorted descending with a
STABLE sort (Vec::sort_by).  Only the ordering rule is mirrored; CheckM file parsing is out of scope.
"""
from __future__ import annotations

import numpy as np


def parks2020_reduced_score(completeness, contamination, num_contigs, num_ambiguous_bases) -> np.ndarray:
    comp = np.asarray(completeness, dtype=np.float32).astype(np.float64)
    cont = np.asarray(contamination, dtype=np.float32).astype(np.float64)
    return comp * 100. - 5. * cont * 100. - 5. * np.asarray(num_contigs, dtype=np.float64) / 100. \
        - 5. * np.asarray(num_ambiguous_bases, dtype=np.float64) / 100000.


def quality_order_parks2020_reduced(completeness, contamination, num_contigs, num_ambiguous_bases) -> np.ndarray:
    """Indices of the genomes, best first (stable for equal scores, like Vec::sort_by)."""
    score = parks2020_reduced_score(completeness, contamination, num_contigs, num_ambiguous_bases)
    if np.any(np.isnan(score)):
        raise ArithmeticError("Arithmetic error while calculating genome quality")  # :1088-1089
    return np.argsort(-score, kind="stable")
