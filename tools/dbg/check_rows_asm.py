"""k_schur_rows keeps one quad of MFMA operands in flight ACROSS inline-asm statements (kernels.hip, SVIN_ROWS_EIGHT): the
compiler does not know that the LDS writes those registers after the statement that requested them has ended.  This script reads
the kernel's disassembly and checks that no instruction outside the statements touches the in-flight registers between the first
request of a row and the drain behind its loop.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics --cuda-device-only -S svin_amd/csrc/kernels.hip -o /tmp/k.s
  python tools/dbg/check_rows_asm.py /tmp/k.s
"""
import re, sys

src = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(src) if re.match(r"^_ZN4svin12k_schur_rowsILi\d+EEEvNS_13DeviceProblemE:", l))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end]

def regs(tok):
    """v12 -> {12}; v[12:13] -> {12, 13}"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out

# statements: (first line, last line, text)
stmts, i = [], 0
while i < len(body):
    if "#ASMSTART" in body[i]:
        j = i
        while "#ASMEND" not in body[j]:
            j += 1
        stmts.append((i, j, body[i + 1:j]))
        i = j
    i += 1
eights = [s for s in stmts if sum("v_mfma" in l for l in s[2]) == 8]
firsts = [s for s in stmts if sum("ds_read_b64" in l for l in s[2]) == 6 and not any("v_mfma" in l for l in s[2])]
drains = [s for s in stmts if len(s[2]) == 1 and "s_waitcnt lgkmcnt(0)" in s[2][0]]
print("%d eight-word statements, %d first-quad statements, %d drains" % (len(eights), len(firsts), len(drains)))
assert eights and firsts and drains
inflight = set()
for s in eights:
    mf = [l for l in s[2] if "v_mfma" in l][:4]          # products of words 0..3: their A / B operands are set A
    for l in mf:
        ops = l.split(None, 1)[1].split(",")
        inflight |= regs(ops[1]) | regs(ops[2])
    # set A is re-requested by the six reads between the fourth and the fifth product: same registers
    k4 = [n for n, l in enumerate(s[2]) if "v_mfma" in l]
    rd = [l for l in s[2][k4[3]:k4[4]] if "ds_read_b64" in l]
    assert len(rd) == 6
    again = set()
    for l in rd:
        again |= regs(l.split(None, 1)[1].split(",")[0])
    assert again == set().union(*[regs(l.split(None, 1)[1].split(",")[1]) | regs(l.split(None, 1)[1].split(",")[2]) for l in mf]), "set A differs"
for s in firsts:
    for l in s[2]:
        if "ds_read_b64" in l:
            inflight |= regs(l.split(None, 1)[1].split(",")[0])
print("registers in flight between statements:", sorted(inflight))
# walk the control flow from the end of every first-quad statement until a drain statement is reached
labels = {}
for n, l in enumerate(body):
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        labels[m.group(1)] = n
stmt_at = {}
for a, b, t in stmts:
    for n in range(a, b + 1):
        stmt_at[n] = (a, b, t)
drain_starts = {a for a, b, t in drains}
bad, seen = 0, set()
work = [f[1] + 1 for f in firsts]
while work:
    n = work.pop()
    while n < len(body) and n not in seen:
        seen.add(n)
        if n in stmt_at:
            a, b, t = stmt_at[n]
            if a in drain_starts:
                break                      # the row is done: its operands have landed
            n = b + 1
            continue
        l = body[n].split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            n += 1
            continue
        if regs(l) & inflight:
            print("line %d touches an in-flight register: %s" % (n, l))
            bad += 1
        m = re.match(r"^(s_branch|s_cbranch_\w+)\s+(\.LBB[0-9_]+)", l)
        if m:
            work.append(labels[m.group(2)])
            if m.group(1) == "s_branch":
                break
        if l.startswith("s_endpgm"):
            break
        n += 1
print("%d instructions between the statements walked" % len([n for n in seen if n not in stmt_at]))
print("OK" if not bad else "%d offending instructions" % bad)
sys.exit(1 if bad else 0)
