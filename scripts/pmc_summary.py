"""Summarise the two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs, each
with --kernel-trace only) into per-kernel HBM traffic per launch.

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads
(TCC_EA0_RDREQ x 64 B with 128-B requests tallied at 64 B), so the read side is doubled; WRITE_SIZE
is taken as reported (uncalibrated)."""
import collections
import csv
import json
import sys


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        d[n][0] += 1
        d[n][1] += float(r["Counter_Value"])
    return d


def main(fetch_csv, write_csv, out_json):
    f, w = agg(fetch_csv, "FETCH_SIZE"), agg(write_csv, "WRITE_SIZE")
    out = {"_units": "bytes per launch; fetch_bytes = 2 * FETCH_SIZE[KiB] * 1024 (gfx950 correction), "
                     "write_bytes = WRITE_SIZE[KiB] * 1024"}
    for n in sorted(f, key=lambda n: -f[n][1]):
        calls = f[n][0]
        fk = f[n][1] / calls
        wk = w[n][1] / max(w[n][0], 1) if n in w else 0.0
        out[n] = {"launches": calls, "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                  "fetch_bytes": int(2 * fk * 1024), "write_bytes": int(wk * 1024),
                  "hbm_bytes": int(2 * fk * 1024 + wk * 1024)}
    json.dump(out, open(out_json, "w"), indent=1)
    # the dominant kernel family (bench.py's `roofline`): launch-weighted HBM bytes per KPConv gather launch
    kp = {n: v for n, v in out.items() if isinstance(v, dict) and "k_kpconv" in n}
    launches = sum(v["launches"] for v in kp.values())
    if launches:
        json.dump({"source": "%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes with --kernel-trace only, "
                             "bench.py --isolated-only --steps 3 --warmup 1)" % out_json.split("/")[-1],
                   "correction": "fetch bytes = 2 x FETCH_SIZE[KiB] x 1024 (gfx950: FETCH_SIZE reports half of wide coalesced "
                                 "reads); write bytes = WRITE_SIZE[KiB] x 1024",
                   "kernels": kp,
                   "hbm_bytes_per_launch": int(sum(v["hbm_bytes"] * v["launches"] for v in kp.values()) / launches)},
                  open(out_json.replace("_pmc_traffic.json", "_pmc_kpconv.json"), "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
