"""The reference's dump-directory weight format (python side).

Written by /root/reference/python/dump.py:130-210, read by
/root/reference/src/model/load.rs:19-310: every file is a 1-D little-endian
float32 .npy whose contents are [dim_0, ..., dim_{D-1}, v_0, v_1, ...] -- the shape
is stored as floats in front of the data; scalars are [1.0, value].  The native
loader is `wb_model_load_dump_dir` (csrc/model_load.cpp); this module writes
fixtures in that format and reads them back for tests.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np


def write_dump_dir(weights: dict, path: str) -> None:
    for name, arr in weights.items():
        arr = np.asarray(arr, dtype=np.float32)
        f = os.path.join(path, name + ".npy")
        os.makedirs(os.path.dirname(f), exist_ok=True)
        flat = np.concatenate([np.asarray(arr.shape, dtype=np.float32), arr.reshape(-1)])
        np.save(f, flat)


# rank the loader expects for each tensor leaf (load.rs: load_tensor::<B, D>)
def _rank_of(name: str) -> int:
    leaf = name.rsplit("/", 1)[-1]
    if name.endswith("conv1/weight") or name.endswith("conv2/weight"):
        return 3
    if leaf == "positional_embedding" or (leaf == "weight" and "_ln" not in name
                                          and not name.endswith("ln/weight")
                                          and not name.endswith("ln_post/weight")):
        return 2
    return 1


def read_dump_dir(path: str) -> "OrderedDict[str, np.ndarray]":
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for root, _, files in sorted(os.walk(path)):
        for fn in sorted(files):
            if not fn.endswith(".npy"):
                continue
            full = os.path.join(root, fn)
            name = os.path.relpath(full, path)[:-4].replace(os.sep, "/")
            flat = np.load(full).astype(np.float32)
            r = _rank_of(name)
            shape = tuple(int(v) for v in flat[:r])
            out[name] = flat[r:].reshape(shape)
    return out
