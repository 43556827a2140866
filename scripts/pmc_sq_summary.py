"""Per-kernel SQ counter summary of scripts/pmc_sq.sh's passes (rocprofv3 --pmc, csv): sums per launch and the ratios
that say what a kernel's wavefronts spend their cycles on (MI355X_MICROARCH.md, rocprofv3 PMC slots: WAIT_ANY +
WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES in quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles)."""
import collections
import csv
import sys


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        d[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return d, {k: len(v) for k, v in n.items()}


def main(p1, p2):
    a, na = load(p1)
    b, nb = load(p2)
    rows = sorted(a, key=lambda k: -a[k].get("SQ_WAVE_CYCLES", 0))
    print("%-52s %5s %9s | %5s %5s %5s | %5s %5s %5s | %6s %6s" % ("kernel", "calls", "waveMcyc", "act%", "wInst", "wAny",
                                                                    "valu%", "lds%", "mfma%", "ldsCf%", "busy"))
    for k in rows[:28]:
        c = a[k]
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        busy = c["SQ_BUSY_CYCLES"] or 1.0
        d = b.get(k, {})
        print("%-52s %5d %9.2f | %5.1f %5.1f %5.1f | %5.1f %5.1f %5.1f | %6.1f %6.2f" % (
            k[:52], na[k], wc / na[k] / 1e6, 100 * c["SQ_ACTIVE_INST_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc,
            100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_VALU"] / wc, 100 * c["SQ_ACTIVE_INST_LDS"] / wc,
            # MFMA busy is in cycles per SIMD-ish units; relate it to the busy window: 4 SIMDs x CUs are summed by the tool
            100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * busy) if busy else 0.0,
            100 * d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1), 1), busy / na[k] / 1e6))
    print()
    print("%-52s %9s %9s %9s %9s %9s | %7s %7s" % ("kernel (per launch)", "valuInst", "ldsInst", "vmemRd", "saluInst", "waves",
                                                "wLDS%", "ldsAct"))
    for k in rows[:28]:
        d, n = b.get(k, {}), max(nb.get(k, 1), 1)
        wc = a[k]["SQ_WAVE_CYCLES"] or 1.0
        print("%-52s %9.0f %9.0f %9.0f %9.0f %9.0f | %7.1f %7.0f" % (
            k[:52], d.get("SQ_INSTS_VALU", 0) / n, d.get("SQ_INSTS_LDS", 0) / n, d.get("SQ_INSTS_VMEM_RD", 0) / n,
            d.get("SQ_INSTS_SALU", 0) / n, d.get("SQ_WAVES", 0) / n, 100 * d.get("SQ_WAIT_INST_LDS", 0) / (wc * n / na[k]),
            d.get("SQ_LDS_IDX_ACTIVE", 0) / n))
    print("\nact/wInst/wAny: share of wave cycles issuing / stalled at issue / parked (waitcnt, barrier); valu, lds: share of "
          "wave cycles with such an instruction active; mfma%: SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES); ldsCf%: "
          "bank-conflict cycles / LDS active cycles")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
