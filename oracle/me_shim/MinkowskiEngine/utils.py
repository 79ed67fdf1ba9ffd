"""TEST INFRASTRUCTURE ONLY -- MinkowskiEngine.utils stand-in."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import imf_oracle as _o  # noqa: E402


def sparse_quantize(coordinates, features=None, return_index=False, **_):
    c = np.floor(np.asarray(coordinates)).astype(np.int64)
    c4 = np.concatenate([np.zeros((len(c), 1), np.int64), c], 1)
    inds = _o.first_occurrence_unique(_o.pack_keys(c4))
    out = c[inds].astype(np.int32)
    if features is not None:
        return (out, features[inds], inds) if return_index else (out, features[inds])
    return (out, inds) if return_index else out


def batched_coordinates(coords_list, dtype=torch.int32, device=None):
    rows = []
    for b, c in enumerate(coords_list):
        c = torch.as_tensor(np.asarray(c)).to(dtype)
        rows.append(torch.cat([torch.full((len(c), 1), b, dtype=dtype), c], 1))
    return torch.cat(rows, 0)


def fnv_hash_vec(arr):
    return _o.fnv_hash_vec(arr)
