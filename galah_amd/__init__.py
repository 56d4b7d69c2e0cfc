"""galah_amd -- MI355X-native finch-precluster / ANI hot path of wwood/galah.

Host-side mirror of the reference's plugin interface for this path:
  FinchPreclusterer            <- src/finch.rs (PreclusterDistanceFinder)
  HipAniClusterer              <- src/skani.rs SkaniClusterer (ClusterDistanceFinder)
  SortedPairGenomeDistanceCache<- src/sorted_pair_genome_distance_cache.rs
  cluster                      <- src/clusterer.rs::cluster
Everything computes in libgalah_hip.so (include/galah_hip.h); importing the package does not
load the library, using it does -- and raises if the HIP extension has not been built.
"""
from .cache import SortedPairGenomeDistanceCache
from .engine import Context, cluster_pairs, cluster_pairs_lazy, device_count, fasta_stream
from .finch import FinchPreclusterer, distances
from .ani import HipAniClusterer
from .clusterer import GalahClusterer, cluster
from .quality import parks2020_reduced_score, quality_order_parks2020_reduced
from ._lib import GalahHipError, PAIR_DTYPE, get_options, set_options

__all__ = ["SortedPairGenomeDistanceCache", "Context", "cluster_pairs", "cluster_pairs_lazy", "device_count", "fasta_stream", "FinchPreclusterer",
           "distances", "HipAniClusterer", "cluster", "GalahClusterer", "GalahHipError", "PAIR_DTYPE", "parks2020_reduced_score",
           "quality_order_parks2020_reduced", "get_options", "set_options"]
