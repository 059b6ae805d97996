"""Child process of test_debug_build_bounds_checks_page_ids: runs with MILLION_HIP_LIB pointing at the
-DMILLION_DEBUG_CHECK_IDS build of the library (million_amd/libmillion_hip_dbgids.so, `make debug-ids`).

For each of the three decode-attention kernels (streaming MFMA, tile, scalar): a clean paged call counts 0 bad ids and
matches the oracle; the same call with page ids outside the pools (too large, negative) inside the live context is
counted, reads page 0 instead (finite output, no fault), and leaves the heads of the untouched kv head exact.
Prints one JSON line."""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    import torch
    from million_amd import _lib, ops
    from oracle import oracle as O
    from tests import synth

    lib = _lib.load()
    res = {"lib": str(_lib.LIB_PATH.name), "cases": []}
    assert lib.million_debug_bad_page_ids() == 0, "not a MILLION_DEBUG_CHECK_IDS build (or stale counter)"
    cases = [("stream", 128, 64, False, False), ("stream-i64", 128, 64, False, True), ("tile", 64, 32, False, False),
             ("scalar", 128, 64, True, False)]
    for name, d, M, generic, i64 in cases:
        C, ps, T, r, nhk = 256, 64, 1500, 9, 2
        c = synth.attn_case(4100 + d + M, 1, 4 * nhk, nhk, d, M, C, T, r)
        gold = O.decode_attn(**c)
        vpool, ids = O.v_rowmajor_to_pool(c["v_codes"], ps)
        kpool, _ = O.k_rowmajor_to_pool(c["k_codes"], ps)
        n_pool = vpool.shape[0]
        t = {k: torch.from_numpy(v).cuda() for k, v in c.items() if isinstance(v, np.ndarray)}
        kp, vp = ops.prepare_cents(t["k_cents"], cache=False), ops.prepare_cents(t["v_cents"], cache=False)
        kc, vc = torch.from_numpy(kpool).cuda(), torch.from_numpy(vpool).cuda()
        dt = np.int64 if i64 else np.int32

        def run(id_arr):
            ids_t = torch.from_numpy(id_arr.astype(dt)).cuda()
            out = ops.pq_decode_attn(t["q"], kc, vc, kp, vp, t["k_res"], t["v_res"], r, M=M, C=C, n_tokens=T,
                                     k_page_ids=ids_t, v_page_ids=ids_t, page_size=ps)
            torch.cuda.synchronize()
            return out.float().cpu().numpy()

        ops.set_force_generic(generic)
        try:
            clean = run(ids)
            n_clean = lib.million_debug_bad_page_ids()
            bad = ids.copy()
            bad[0, 0, 3] = n_pool + 5             # beyond the pool
            bad[0, 0, 11] = -7                    # negative
            bad[0, 0, 20] = 2 ** 30               # far away: an unchecked read of this page faults
            hurt = run(bad)
            n_bad = lib.million_debug_bad_page_ids()
        finally:
            ops.set_force_generic(False)
        G = 4
        res["cases"].append({
            "name": name, "clean_bad_ids": int(n_clean), "bad_ids": int(n_bad),
            "clean_err": float(np.abs(clean - gold).mean()),
            "finite": bool(np.isfinite(hurt).all()),
            "other_head_err": float(np.abs(hurt[:, G:] - gold[:, G:]).mean()),      # kv head 1 kept its ids
            "hurt_head_moved": bool(np.abs(hurt[:, :G] - gold[:, :G]).max() > 0),
        })
    print(json.dumps(res))


if __name__ == "__main__":
    main()
