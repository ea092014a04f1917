"""Marker dictionaries -- host-side mirror of `ARDictionary` (src/dictionaries.rs:21-232).

The 14 code tables of src/dictionaries.rs:5-19 live in data/dictionaries.bin (one
little-endian u64 blob, produced by tools/extract_dictionaries.py) with the
name -> {num_bits, tau, offset, count} index of src/dictionaries.rs:30-113 in
data/dictionaries.json.  This class only holds the table; every computation on it
(nearest-code search, tau for the tables that declare tau == 0) runs on the GPU through
the C ABI (a3_find_nearest / a3_calculate_tau) and fails loudly without it.
"""
import json
import math
from pathlib import Path

import numpy as np

_DATA = Path(__file__).resolve().parent / "data"
_INDEX = None
_BLOB = None


def _load():
    global _INDEX, _BLOB
    if _INDEX is None:
        _INDEX = json.loads((_DATA / "dictionaries.json").read_text())
        _BLOB = np.fromfile(_DATA / "dictionaries.bin", dtype="<u8")
    return _INDEX, _BLOB


class ARDictionary:
    """`ARDictionary { num_bits, tau, code_list }` (src/dictionaries.rs:22-27)."""

    def __init__(self, num_bits: int, tau: int, code_list: np.ndarray, name: str = ""):
        self.num_bits = int(num_bits)
        self._tau = int(tau)
        self.code_list = np.ascontiguousarray(code_list, dtype=np.uint64)
        self.name = name

    # src/dictionaries.rs:140-145 -- unknown names panic in the reference; here they raise.
    @classmethod
    def new_from_named_dict(cls, code_name: str) -> "ARDictionary":
        index, blob = _load()
        key = code_name.upper()
        if key not in index:
            raise KeyError("TODO: code for this dict is not implemented.")
        e = index[key]
        return cls(e["num_bits"], e["tau"], blob[e["offset"]: e["offset"] + e["count"]].copy(), key)

    # src/dictionaries.rs:147-149
    @staticmethod
    def get_dictionary_names():
        index, _ = _load()
        return sorted(index.keys())

    # src/dictionaries.rs:116-127: tau == 0 in the table means "min pairwise distance";
    # computed on first use by the device kernel behind a3_calculate_tau.
    @property
    def tau(self) -> int:
        if self._tau == 0:
            from . import _lib

            self._tau = _lib.calculate_tau(self.code_list)
        return self._tau

    # src/dictionaries.rs:154-156
    def get_mark_size(self) -> int:
        return int(math.ceil(math.sqrt(float(self.num_bits)))) + 2

    # src/dictionaries.rs:160-196 (device kernel; lowest index wins ties)
    def find_nearest(self, bits: int):
        from . import _lib

        idx, dist = _lib.find_nearest(self.code_list, np.array([bits], dtype=np.uint64))
        return int(idx[0]), int(dist[0])

    # src/dictionaries.rs:200-207
    def try_find_nearest(self, bits: int):
        idx, dist = self.find_nearest(bits)
        return (idx, dist) if dist < self.tau else None

    # src/dictionaries.rs:212-232 -- pure bit unpacking, LSB-first
    def make_binary_image(self, marker_id: int):
        code = int(self.code_list[marker_id])
        width = self.get_mark_size()
        bits = [False] * width
        for i in range(self.num_bits):
            if (len(bits) & 0xFF) % width == 0:
                bits.append(False)
            bits.append(code & (1 << i) != 0)
            if (len(bits) & 0xFF) % width == width - 1:
                bits.append(False)
        bits.extend([False] * width)
        return width, bits
