"""Static check of the hand-scheduled kernels in an assembly listing (hipcc -S --cuda-device-only): between a
global_load and the s_waitcnt that covers it, no instruction may read or write the load's destination registers.
The function body is walked in layout order, its largest loop twice (for the registers that live across the loop
edge); loads return in order, stores are counted as outstanding entries without registers; scratch (spill) and flat
accesses count against the same counter and are walked as entries too.
usage: python tools/check_pending_regs.py file.s <mangled-name-fragment> [...]"""
import re, sys


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(lines, name):
    pend, bad, nload = [], 0, 0
    for l in lines:
        l = l.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        op = l.split()[0]
        m = re.search(r"vmcnt\((\d+)\)", l) if op == "s_waitcnt" else None
        if op == "s_waitcnt":
            if m:
                del pend[:max(0, len(pend) - int(m.group(1)))]
            continue
        used = regs(l)
        for p in pend:
            if p & used:
                bad += 1
                print("%s: '%s' touches registers of a load in flight %s" % (name, l, sorted(p & used)))
        # everything that counts against vmcnt is an entry: spill traffic (scratch_*) and flat accesses too, or a
        # reload the compiler places inside a hand-issued batch would make the batch's counted wait one entry short
        if op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")):
            pend.append(regs(l.split(",")[0]))
            nload += 1
        elif op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic", "buffer_atomic",
                            "flat_atomic")):
            pend.append(set())
    print("%s: %d loads walked, %d violations" % (name, nload, bad))
    return bad


def check_file(path, frags):
    """Number of violations over every function of `path` whose name contains one of `frags` (and how many matched)."""
    s = open(path).read()
    total, nfun = 0, 0
    for frag in frags:
        for m in re.finditer(r"^(\S*%s\S*):[^\n]*\n" % re.escape(frag), s, re.M):
            body = s[m.end():s.index("s_endpgm", m.end())].split("\n")
            labels = {mm.group(1): n for n, l in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
            loops = [(labels[mm.group(1)], n) for n, l in enumerate(body)
                     for mm in [re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)] if mm and labels.get(mm.group(1), 1 << 30) < n]
            if loops:
                a, b = max(loops, key=lambda t: t[1] - t[0])
                body = body[:b + 1] + body[a:b + 1] + body[b + 1:]
            total += check(body, m.group(1))
            nfun += 1
    return total, nfun


def scratch_ops(path, frags):
    """Spill instructions (scratch_load / scratch_store) in the bodies of the functions `check_file` would walk."""
    s = open(path).read()
    n = 0
    for frag in frags:
        for m in re.finditer(r"^(\S*%s\S*):[^\n]*\n" % re.escape(frag), s, re.M):
            body = s[m.end():s.index("s_endpgm", m.end())]
            n += len(re.findall(r"^\s*scratch_(?:load|store)", body, re.M))
    return n


if __name__ == "__main__":
    sys.exit(1 if check_file(sys.argv[1], sys.argv[2:])[0] else 0)
