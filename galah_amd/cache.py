"""SortedPairGenomeDistanceCache mirror (reference src/sorted_pair_genome_distance_cache.rs:5-59).

A BTreeMap<(usize, usize), Option<f32>> whose key is sorted on insert/get.  Stored as a dict plus a
lazily sorted key list; iteration order equals the BTreeMap's.  Values are numpy float32 (or None),
so equality checks are on the exact f32 bits the reference stores.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

Key = Tuple[int, int]


def _sorted_key(ids: Key) -> Key:
    # cache.rs:22-28: `if ids.0 < ids.1 {(0,1)} else {(1,0)}`
    return (ids[0], ids[1]) if ids[0] < ids[1] else (ids[1], ids[0])


class SortedPairGenomeDistanceCache:
    def __init__(self):
        self._dict: Optional[Dict[Key, Optional[np.float32]]] = {}
        self._src: Optional[np.ndarray] = None   # from_pairs: the back-end's sorted edge list; the map is built on first use

    @property
    def _d(self) -> Dict[Key, Optional[np.float32]]:
        if self._dict is None:
            p = self._src
            lo = np.minimum(p["i"], p["j"]).tolist()
            hi = np.maximum(p["i"], p["j"]).tolist()
            self._dict = dict(zip(zip(lo, hi), list(p["ani"].astype(np.float32))))
        return self._dict

    def insert(self, genome_ids: Key, distance: Optional[float]) -> None:
        self._d[_sorted_key(genome_ids)] = None if distance is None else np.float32(distance)

    def get(self, genome_ids: Key):
        """Returns None when the key is absent, else a 1-tuple holding Option<f32> (None or float32)."""
        k = _sorted_key(genome_ids)
        if k in self._d:
            return (self._d[k],)
        return None

    def contains_key(self, genome_ids: Key) -> bool:
        return _sorted_key(genome_ids) in self._d

    def transform_ids(self, input_ids: Sequence[int]) -> "SortedPairGenomeDistanceCache":
        out = SortedPairGenomeDistanceCache()
        for i, g1 in enumerate(input_ids):
            for j in range(i + 1, len(input_ids)):
                got = self.get((g1, input_ids[j]))
                if got is not None:
                    out.insert((i, j), got[0])
        return out

    def __len__(self) -> int:
        return len(self._src) if self._dict is None else len(self._dict)

    def items(self) -> List[Tuple[Key, Optional[np.float32]]]:
        return sorted(self._d.items())

    def __iter__(self) -> Iterator[Key]:
        return iter(sorted(self._d))

    def __eq__(self, other) -> bool:
        if not isinstance(other, SortedPairGenomeDistanceCache):
            return NotImplemented
        if self._d.keys() != other._d.keys():
            return False
        for k, v in self._d.items():
            w = other._d[k]
            if (v is None) != (w is None):
                return False
            if v is not None and np.float32(v).tobytes() != np.float32(w).tobytes():
                return False
        return True

    def __repr__(self) -> str:
        # Debug format of the reference: SortedPairGenomeDistanceCache { internal: {(0, 1): Some(0.99)} }
        def fmt(v):
            return "None" if v is None else f"Some({_rust_f32(v)})"
        body = ", ".join(f"({a}, {b}): {fmt(v)}" for (a, b), v in self.items())
        return f"SortedPairGenomeDistanceCache {{ internal: {{{body}}} }}"

    @staticmethod
    def from_pairs(pairs: np.ndarray) -> "SortedPairGenomeDistanceCache":
        """From a back-end's edge list (unique unordered pairs).  The map itself is only built when something asks
        for it: galah_amd.cluster hands the edge list straight to the host clusterer, and a Python dict of 45 000
        tuple keys costs more than the whole GPU pipeline."""
        c = SortedPairGenomeDistanceCache()
        c._src = np.array(pairs, copy=True)
        c._dict = None
        return c


def _rust_f32(v) -> str:
    """Shortest decimal that round-trips the f32 (Rust's {:?} for f32)."""
    s = np.format_float_positional(np.float32(v), unique=True, trim="0")
    return s if "." in s else s + ".0"
