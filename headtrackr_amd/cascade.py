"""Reader for the HTCB cascade blob (format: headtrackr_amd/js/cascade_pack.js).

The blob holds the trained ccv BBF face cascade that the reference exposes as ``headtrackr.cascade``
(reference: src/cascade.js:19).  ``headtrackr_amd/data/cascade.bin`` is produced by ``tools/pack_cascade.js``.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass

import numpy as np

MAXPTS = 8
_HEADER = 32
_STAGE = np.dtype([("count", "<u4"), ("first", "<u4"), ("threshold", "<f8")])
_FEAT = np.dtype(
    [
        ("size", "u1"),
        ("pad", "u1", (7,)),
        ("px", "i1", (MAXPTS,)),
        ("py", "i1", (MAXPTS,)),
        ("pz", "i1", (MAXPTS,)),
        ("nx", "i1", (MAXPTS,)),
        ("ny", "i1", (MAXPTS,)),
        ("nz", "i1", (MAXPTS,)),
        ("alpha", "<f8", (2,)),
    ]
)
assert _STAGE.itemsize == 16 and _FEAT.itemsize == 72

DEFAULT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cascade.bin")


@dataclass
class Cascade:
    width: int
    height: int
    stages: np.ndarray  # structured (_STAGE), one row per stage
    features: np.ndarray  # structured (_FEAT), all stages concatenated
    blob: bytes

    @property
    def count(self) -> int:
        return int(self.stages.shape[0])


def load_cascade(path: str | None = None) -> Cascade:
    with open(path or DEFAULT_PATH, "rb") as f:
        blob = f.read()
    return parse_cascade(blob)


def parse_cascade(blob: bytes) -> Cascade:
    magic, version, nst, w, h, nfeat, maxpts, _ = struct.unpack_from("<4sIIIIIII", blob, 0)
    if magic != b"HTCB" or version != 1 or maxpts != MAXPTS:
        raise ValueError("not an HTCB v1 cascade blob")
    stages = np.frombuffer(blob, dtype=_STAGE, count=nst, offset=_HEADER)
    feats = np.frombuffer(blob, dtype=_FEAT, count=nfeat, offset=_HEADER + nst * _STAGE.itemsize)
    if len(blob) != _HEADER + nst * _STAGE.itemsize + nfeat * _FEAT.itemsize:
        raise ValueError("HTCB blob has trailing or missing bytes")
    return Cascade(width=w, height=h, stages=stages, features=feats, blob=bytes(blob))
