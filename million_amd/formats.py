"""On-disk formats of the reference's offline stages, so that trained codebooks and sampled
K/V vectors made by the reference can be plugged into this library unchanged.

* centroids `key_cent_{M}_{nbits}.pq.pt` / `val_cent_{M}_{nbits}.pq.pt`: `torch.save` of an fp32 tensor
  `(M, 2**nbits, d // M)` (reference scripts/modeldb/main_pq.py:222-237, written from faiss by
  scripts/utils/pq_utils.py:586-609; read at main_pq.py:257-260 with `weights_only=True` and cast to the model
  dtype).  Loaded here with `weights_only=True` only: nothing from the file is executed.
* `.fvecs` (scripts/utils/fvecio.py:23-43): per vector a little-endian int32 dimension `d` followed by `d`
  float32 values; the writer appends (`mode='ab'`) and creates the file when it does not exist.
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Tuple

import numpy as np
import torch


def centroid_paths(cent_root, M: int, nbits: int) -> Tuple[Path, Path]:
    """File names the reference uses under `centroids/<model>/<dataset>/` (main_pq.py:116,225,236)."""
    root = Path(cent_root)
    return root / f"key_cent_{M}_{nbits}.pq.pt", root / f"val_cent_{M}_{nbits}.pq.pt"


def load_centroid_file(path, *, M: int = None, nbits: int = None, d: int = None, dtype=torch.float16,
                       device="cpu") -> torch.Tensor:
    t = torch.load(Path(path), map_location="cpu", weights_only=True)
    if not isinstance(t, torch.Tensor) or t.dim() != 3:
        raise ValueError(f"{path}: expected one (M, 2**nbits, d/M) tensor, got {type(t).__name__}"
                         + (f" of shape {tuple(t.shape)}" if isinstance(t, torch.Tensor) else ""))
    if M is not None and t.shape[0] != M:
        raise ValueError(f"{path}: M={t.shape[0]}, expected {M}")
    if nbits is not None and t.shape[1] != 2 ** nbits:
        raise ValueError(f"{path}: {t.shape[1]} centroids, expected 2**{nbits}")
    if d is not None and t.shape[0] * t.shape[2] != d:
        raise ValueError(f"{path}: M * d_m = {t.shape[0] * t.shape[2]}, expected d={d}")
    return t.to(dtype).to(device).contiguous()


def load_centroids(cent_root, M: int, nbits: int, *, d: int = None, dtype=torch.float16, device="cpu"):
    """(key_cent, value_cent) as `set_cent` wants them (pq_utils.py:149-159)."""
    kp, vp = centroid_paths(cent_root, M, nbits)
    return (load_centroid_file(kp, M=M, nbits=nbits, d=d, dtype=dtype, device=device),
            load_centroid_file(vp, M=M, nbits=nbits, d=d, dtype=dtype, device=device))


def save_centroids(cent_root, key_cent: torch.Tensor, value_cent: torch.Tensor, nbits: int = 8) -> Tuple[Path, Path]:
    """Write the pair the way the reference's training stage does (fp32, main_pq.py:224-237)."""
    M = key_cent.shape[0]
    for c in (key_cent, value_cent):
        if c.dim() != 3 or c.shape[1] != 2 ** nbits or c.shape[0] != M:
            raise ValueError("centroids must be (M, 2**nbits, d/M)")
    os.makedirs(cent_root, exist_ok=True)
    kp, vp = centroid_paths(cent_root, M, nbits)
    torch.save(key_cent.detach().float().cpu().contiguous(), kp)
    torch.save(value_cent.detach().float().cpu().contiguous(), vp)
    return kp, vp


def read_fvecs(filename) -> np.ndarray:
    """All vectors of an .fvecs file as (n, d) float32.  Same result as the reference's record loop
    (fvecio.py:23-33) for files whose vectors share one dimension - which is every file its writer produces from one
    sampler; a trailing fragment shorter than a header is ignored like there."""
    raw = np.fromfile(filename, dtype=np.uint8)
    if raw.size < 4:
        return np.zeros((0, 0), dtype=np.float32)
    d = int(raw[:4].view("<i4")[0])
    if d <= 0:
        raise ValueError(f"{filename}: bad vector dimension {d}")
    rec = 4 + 4 * d
    n = raw.size // rec
    if raw.size - n * rec >= 4:      # another header follows: a record of a different dimension, or a truncated one
        raise ValueError(f"{filename}: trailing bytes do not form a {d}-dimensional record (mixed dimensions or truncated file)")
    body = raw[: n * rec].reshape(n, rec)
    dims = body[:, :4].copy().view("<i4").ravel()
    if not (dims == d).all():
        raise ValueError(f"{filename}: vectors of different dimensions ({sorted(set(dims.tolist()))[:4]}...)")
    return body[:, 4:].copy().view("<f4").reshape(n, d)


def write_fvecs(filename, vecs, mode: str = "ab") -> None:
    """Append (default) or write vectors; creates the file when appending to a missing one (fvecio.py:35-43)."""
    if mode == "ab" and not os.path.exists(filename):
        mode = "wb"
    vecs = np.asarray(vecs)
    if vecs.ndim == 1:
        vecs = vecs[None]
    n, d = vecs.shape
    rec = np.empty((n, 1 + d), dtype="<f4")
    rec[:, 0] = np.array([d], dtype="<i4").view("<f4")[0]
    rec[:, 1:] = vecs.astype("<f4")
    with open(filename, mode) as f:
        f.write(rec.tobytes())
