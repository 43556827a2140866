"""Generate tests/golden/frontend_*.{npz,json} from the UNMODIFIED reference C++ front end
(oracle/_ref/libpcrcg_ref.so, built by oracle/Makefile from /root/reference/cpp_wrappers.zip).

Runs only in the build container.  Inputs are reproducible from (recipe, seed) through
pcrcg_amd.synthetic; outputs are stored in full for the `mini` recipe and as SHA-256 digests for
C1 / S30k (neighbour tables are digested after canonicalising the order inside groups of exactly
equal squared distance, whose reference order is arbitrary -- SURVEY.md 8a-2).
"""
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import frontend as F  # noqa: E402
from pcrcg_amd import synthetic as S  # noqa: E402
from tests.tieutil import canonicalise_table  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pyramid(recipe, seed, r0=0.0625, dl0=0.05, levels=4):
    src, tgt = S.pair(recipe, seed)
    pts = np.concatenate([src, tgt])
    lens = np.array([len(src), len(tgt)], np.int32)
    r, dl = r0, dl0
    out = {}
    for l in range(levels):
        out[f"points{l}"] = pts
        out[f"lens{l}"] = lens
        out[f"conv{l}"] = F.ref_batch_query(pts, pts, lens, lens, r)
        if l == levels - 1:
            break
        sp, sl = F.ref_subsample_batch(pts, lens, dl)
        out[f"pool{l}"] = F.ref_batch_query(sp, pts, sl, lens, r)
        out[f"up{l}"] = F.ref_batch_query(pts, sp, lens, sl, 2 * r)
        pts, lens = sp, sl
        r *= 2
        dl *= 2
    return out


def main():
    F.build(ref=True)
    os.makedirs(OUT, exist_ok=True)
    mini = pyramid("mini", 0)
    np.savez_compressed(os.path.join(OUT, "frontend_mini.npz"),
                        **{k: v for k, v in mini.items() if not k.startswith("points0")})
    path = os.path.join(OUT, "frontend_digests.json")
    digests = json.load(open(path)) if os.path.exists(path) else {}      # keeps T8k (scripts/make_golden_ties.py)
    for recipe in ("C1", "S30k"):
        p = pyramid(recipe, 0)
        d = {}
        for k, v in p.items():
            if k.startswith(("points", "lens")):
                d[k] = {"shape": list(v.shape), "sha256": sha(v)}
            else:
                l = int(k[-1])
                if k.startswith("conv"):
                    q, s = p[f"points{l}"], p[f"points{l}"]
                elif k.startswith("pool"):
                    q, s = p[f"points{l + 1}"], p[f"points{l}"]
                else:
                    q, s = p[f"points{l}"], p[f"points{l + 1}"]
                canon, tie_rows = canonicalise_table(v, q, s)
                # sha256: the table exactly as the reference returns it (its own order inside tie groups)
                d[k] = {"shape": list(v.shape), "sha256": sha(v), "sha256_canonical": sha(canon),
                        "tie_rows": int(tie_rows)}
        digests[recipe] = d
    with open(path, "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)
    # libstdc++ unordered_map iteration orders for a few key sets (pins oracle_umap_order and the HIP path)
    rng = np.random.RandomState(7)
    um = {}
    for n in (1, 2, 13, 14, 29, 30, 500, 6000):
        k = np.unique(rng.randint(0, 1 << 40, size=3 * n).astype(np.uint64))
        rng.shuffle(k)
        k = k[:n]
        um[f"keys{n}"] = k
        um[f"order{n}"] = F.ref_umap_order(k)
    np.savez_compressed(os.path.join(OUT, "umap_order.npz"), **um)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
