"""Static check of the GEMM kernels' ISA for the one hazard hipcc cannot see (csrc/gemm_x6.hip loads its operand tiles with
inline-asm `global_load_dword[x4]` and counts `s_waitcnt vmcnt(N)` by hand): any instruction that READS a register a load is
still writing -- typically a `v_mov` the register allocator places in front of the wait when it decides to keep the loaded
value somewhere else.  The result is garbage operands in a kernel that compiles and mostly runs.

Model: walk every kernel's instructions in program order, block by block along fall-through and branch edges (a worklist
over basic blocks, state = the queue of in-flight asm loads as (dest registers) oldest first).  `s_waitcnt vmcnt(N)` retires
all but the N newest.  An instruction that reads or writes a register of an in-flight load is reported.  Only asm-issued
loads are tracked (they sit between ;;#ASMSTART / ;;#ASMEND); the compiler's own loads carry the compiler's own waits.

usage: python scripts/isa_hazards.py file.s  -> prints findings, exit code 1 if any."""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse_kernels(lines):
    """-> {name: [(label or None, instruction text, in_asm)]}"""
    kernels, cur, name, in_asm = {}, None, None, False
    for raw in lines:
        line = raw.split(";;#")[0] if ";;#" not in raw[:6] else raw
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", raw)
        if m:
            cur.append((m.group(1), None, False))
            continue
        s = s.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        cur.append((None, s, in_asm))
        if s.startswith("s_endpgm"):
            cur = None
    return kernels


def check_kernel(name, ins):
    label_at = {lab: i for i, (lab, _, _) in enumerate(ins) if lab}
    findings = []
    seen_states = {}
    work = [(0, ())]
    while work:
        i, inflight = work.pop()
        while i < len(ins):
            key = (i, inflight)
            lab, text, in_asm = ins[i]
            if lab is not None:
                if seen_states.get(i) is not None and inflight in seen_states[i]:
                    break
                seen_states.setdefault(i, set()).add(inflight)
                i += 1
                continue
            op = text.split()[0]
            if in_asm and op.startswith("global_load"):
                dst = text.split()[1].rstrip(",")
                addr_regs = regs(text.split(",", 1)[1])
                live = set().union(*[set(d) for d in inflight]) if inflight else set()
                if addr_regs & live:
                    findings.append((name, i, text, "address registers still being loaded"))
                inflight = inflight + (tuple(sorted(regs(dst))),)
                i += 1
                continue
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", text)
            if m:
                n = int(m.group(1))
                inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight
                if n == 0:
                    inflight = ()
                i += 1
                continue
            if op.startswith("s_waitcnt") and "vmcnt" not in text:
                i += 1
                continue
            if inflight:
                live = set().union(*[set(d) for d in inflight])
                touched = regs(text) & live
                if touched and not op.startswith("s_"):
                    findings.append((name, i, text, "touches v%s while its load is in flight" % sorted(touched)[:4]))
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = text.split()[-1]
                if tgt in label_at:
                    j = label_at[tgt]
                    if not (seen_states.get(j) and inflight in seen_states[j]):
                        work.append((j, inflight))
                if op == "s_branch":
                    break
            if op.startswith("s_endpgm"):
                break
            i += 1
    return findings


def main(path, only=None):
    kernels = parse_kernels(open(path).read().splitlines())
    total, checked = [], 0
    for name, ins in kernels.items():
        if only and only not in name:
            continue
        if not any(a and t and t.startswith("global_load") for _, t, a in ins):
            continue
        checked += 1
        total += check_kernel(name, ins)
    seen = set()
    for name, i, text, why in total:
        k = (name, text)
        if k in seen:
            continue
        seen.add(k)
        print("%s\n    #%d  %s    <- %s" % (name[:110], i, text, why))
    print("isa_hazards: %d kernels with asm loads checked, %d finding(s)" % (checked, len(seen)))
    return 1 if seen else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None))
