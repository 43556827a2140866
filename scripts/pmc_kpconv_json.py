"""profiles/<round>_pmc_traffic.json (scripts/pmc_summary.py output) -> profiles/r01_pmc_kpconv.json:
the KPConv gather kernels' HBM bytes per launch (launch-weighted mean over the kernel variants), which
bench.py reports as roofline.traffic."""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
ks = {k: v for k, v in d.items() if k.startswith("pcrcg::k_kpconv_") and "bwd" not in k and isinstance(v, dict)}
launches = sum(v["launches"] for v in ks.values())
total = sum(v["launches"] * v["hbm_bytes"] for v in ks.values())
json.dump({"source": f"{src} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only, "
                     "bench.py --steps 3 --warmup 1)",
           "correction": "fetch bytes = 2 x FETCH_SIZE[KiB] x 1024 (gfx950: FETCH_SIZE reports half of wide coalesced "
                         "reads); write bytes = WRITE_SIZE[KiB] x 1024",
           "kernels": ks, "hbm_bytes_per_launch": int(total / max(launches, 1))}, open(dst, "w"), indent=1)
print(json.load(open(dst))["hbm_bytes_per_launch"], launches)
