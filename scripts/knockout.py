"""Knock-out accounting of the forward's kernel families: what does the pair engine's throughput
gain when a family of launches simply does not happen?  Results are wrong, timing is meaningful -- it tells which
fusion is worth building before it is built (round 2: the row-positive pass +4 %, the normalisation passes +11 %, the
finishing launches of tall statistics +5.6 % that no rewrite could realise, the concat copies +0.3 %).

    python scripts/knockout.py build      (build container or GPU box: writes scripts/micro/libpcrcg_hip_knock.so)
    python scripts/knockout.py run        (GPU box: bench.py once per family with PCRCG_KNOCK=<family>)

The knock-out library is the normal library with runner.hip patched so that the wrapped calls return PCRCG_OK without
launching when their token is in $PCRCG_KNOCK.  Index-producing kernels are never knocked out (garbage indices would
fault).  The runs load it by assigning pcrcg_amd._lib.LIB_PATH before bench.py starts (the product loader itself takes
no redirection); bench.py's line records the path and hash of the library it ran."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "pcrcg_amd", "csrc")
LIB = os.path.join(REPO, "scripts", "micro", "libpcrcg_hip_knock.so")   # a wrong-results measurement build: never next to the product library
FAMILIES = [("pcrcg_copy2d(", "K_COPY"), ("pcrcg_gather_max(", "K_GMAX"), ("pcrcg_instnorm_colsums(", "K_CSUM"),
            ("pcrcg_instnorm_stats_from_partials(", "K_CFIN"), ("pcrcg_instnorm_apply_sums(", "K_APPLY"),
            ("pcrcg_instnorm_apply(", "K_APPLY"), ("instnorm_apply_pack(t.p[g]", "K_PACK"), ("pcrcg_attention(q.p[g]", "K_ATT"),
            ("pcrcg_edgeconv_reduce_sums(", "K_EDGE"), ("kpconv_aggregate_rows(q[g], nq[g]", "K_GATH"),
            ("pcrcg_kpconv_aggregate(q[g], nq[g]", "K_GATH"), ("pcrcg_knn(coords[g]", "K_KNN")]


def build():
    src = open(os.path.join(CSRC, "runner.hip")).read()
    src = src.replace("namespace pcrcg {\n// gemm.hip", '#include <cstring>\nthread_local int pcrcg_knock_forward_count = 0;\nstatic bool ko(const char* name) { static const char* e = '
                      'getenv("PCRCG_KNOCK"); static const int every = getenv("PCRCG_KNOCK_EVERY") ? atoi(getenv("PCRCG_KNOCK_EVERY")) : 1; '
                      'return e && strstr(e, name) && (every <= 1 || (pcrcg_knock_forward_count % every) == 1); }\n'
                      'namespace pcrcg {\n// gemm.hip', 1)
    pat = "    PCRCG_PROPAGATE(validate_group(model, batches, n));\n    PCRCG_CHECK_ARG(outs && ws);"
    assert pat in src
    src = src.replace(pat, "    ++pcrcg_knock_forward_count;\n" + pat, 1)
    for pat, tok in FAMILIES:
        n = src.count(pat)
        src = src.replace(pat, 'ko("%s") ? PCRCG_OK : %s' % (tok, pat))
        print("%-8s %d call sites" % (tok, n))
    # the KPConv contraction alone (the GEMM a one-kernel KPConv would absorb)
    pat = "            linear(c, wf, blk.kp_wt, kk, nullptr, y, st, inv_n);"
    assert pat in src
    src = src.replace(pat, '            if (!ko("K_KPGEMM")) linear(c, wf, blk.kp_wt, kk, nullptr, y, st, inv_n);')
    tmp = os.path.join(CSRC, "build", "knock")
    os.makedirs(tmp, exist_ok=True)
    open(os.path.join(tmp, "runner_knock.hip"), "w").write(src)
    # GEMMs by row count: PCRCG_KNOCK_M="lo,hi" skips every split-bf16 product with lo <= M < hi
    g = open(os.path.join(CSRC, "gemm_x6.hip")).read()
    pat = "    // k-major operands (a_kmajor: A stored [K, M]; b_kmajor: B stored [K, N]) are read with 4-byte loads"
    assert pat in g
    # PCRCG_KNOCK_M="lo,hi" skips the products with lo <= M < hi; "lo,hi,2" skips them in every SECOND forward of a host
    # thread only: what launching them once for two stacked pairs could buy at most
    g = g.replace(pat, '    { static const char* e = getenv("PCRCG_KNOCK_M"); int lo = 0, hi = 0, every = 1;\n'
                       '      if (e && sscanf(e, "%d,%d,%d", &lo, &hi, &every) >= 2 && m >= lo && m < hi &&\n'
                       '          (every <= 1 || (pcrcg_knock_forward_count % every) == 1)) return PCRCG_OK; }\n' + pat, 1)
    g = g.replace("namespace pcrcg {\nnamespace {\n\ntypedef float f32x16", "extern thread_local int pcrcg_knock_forward_count;\nnamespace pcrcg {\nnamespace {\n\ntypedef float f32x16", 1)
    assert "pcrcg_knock_forward_count;" in g
    open(os.path.join(tmp, "gemm_x6_knock.hip"), "w").write(g)
    objs = [os.path.join(CSRC, "build", f) for f in os.listdir(os.path.join(CSRC, "build"))
            if f.endswith(".o") and f not in ("runner.o", "gemm_x6.o")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + CSRC]
    mine = []
    for name in ("runner_knock", "gemm_x6_knock"):
        subprocess.check_call(["hipcc", *flags, "-c", os.path.join(tmp, name + ".hip"), "-o", os.path.join(tmp, name + ".o")])
        mine.append(os.path.join(tmp, name + ".o"))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *mine, *objs])
    print("built", LIB)


def run():
    import json
    toks = ["none"] + sorted({t for _, t in FAMILIES}) + ["K_KPGEMM", "M:0,1000", "M:1000,5000", "M:5000,20000", "M:20000,100000",
                                                           "M:0,100000", "none"]
    if len(sys.argv) > 2:
        toks = sys.argv[2:]
    for t in toks:
        env = dict(os.environ, PCRCG_KNOCK=t)
        env.pop("PCRCG_KNOCK_M", None)
        env.pop("PCRCG_KNOCK_EVERY", None)
        if t.startswith("M:"):
            env["PCRCG_KNOCK_M"] = t[2:]
        if t.endswith("/2"):                       # "K_EDGE,K_ATT/2": the families, in every second forward call only
            env["PCRCG_KNOCK"], env["PCRCG_KNOCK_EVERY"] = t[:-2], "2"
        boot = ("import sys, runpy; sys.path.insert(0, %r); import pcrcg_amd._lib as L; L.LIB_PATH = %r; "
                "sys.argv = ['bench.py', '--no-cpu-baseline', '--no-extras', '--steps', '100', '--repeats', '3']; runpy.run_path(%r, run_name='__main__')"
                % (REPO, LIB, os.path.join(REPO, "bench.py")))
        out = subprocess.run([sys.executable, "-c", boot], env=env,
                             capture_output=True, text=True).stdout.strip().splitlines()
        print("%-8s %s" % (t, json.loads(out[-1])["value"] if out else "failed"), flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1] if len(sys.argv) > 1 else "build"]()
