"""Build check for the typed-load colour stage (jpeg_kernels.hip, kTyped): the format loads are inline assembly whose
results land asynchronously, so between a load and the `s_waitcnt vmcnt(0)` that precedes the first read the compiler
must not touch the destination registers (no copy at a loop back-edge, no reuse as a temporary, no spill).  This walks the
control-flow graph of every kTyped kernel in the assembly listing with a forward data flow ("registers with a load in
flight", union at joins, cleared by a full vmcnt wait) and FAILS if any instruction names such a register.

    make -C jpeg_amd/csrc asm && python tools/check_typed_loads.py /tmp/jpeg_kernels.s"""
import re
import sys

text = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/jpeg_kernels.s").read()


def regs(l):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", l):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bv(\d+)\b", l))
    return out


total_bad = 0
syms = re.findall(r"^(_ZN\S*encode_image_kernelILb[01]ELb[01]ELi\dELb1E\S*):", text, re.M)
if not syms:
    sys.exit("no kTyped kernel in the listing")
for sym in syms:
    body = text[text.index(sym + ":"):]
    body = body[:body.index(".amdhsa_kernel")]
    lines = [l for l in body.splitlines()[1:]
             if (l.startswith("\t") and not l.strip().startswith((";", "."))) or re.match(r"\.LBB\S+:", l)]
    blocks, cur, label_of = [], {"label": "entry", "ins": []}, {}
    for l in lines:
        m = re.match(r"(\.LBB\S+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        cur["ins"].append(l.strip())
        if re.match(r"s_(c?branch|endpgm)", l.strip()):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    blocks.append(cur)
    blocks = [b for b in blocks if b["ins"] or b["label"]]
    for i, b in enumerate(blocks):
        if b["label"]:
            label_of[b["label"]] = i
    succ = []
    for i, b in enumerate(blocks):
        s, last = [], b["ins"][-1] if b["ins"] else ""
        m = re.match(r"s_c?branch\S*\s+(\.LBB\S+)", last)
        if m:
            s.append(label_of[m.group(1)])
        if not (last.startswith("s_branch") or last.startswith("s_endpgm")) and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    # state = the loads in flight in issue order (tuples of their registers), newest last; a wait vmcnt(N) retires all
    # but the N newest (other memory operations -- stores -- issued in between only retire more).  States are kept
    # per block as a set of alternatives (paths differ at most by what the prologue left in flight).
    state_in = [set() for _ in blocks]
    state_in[0].add(())
    work, seen_bad = [0], {}
    while work:
        i = work.pop()
        outs = set()
        for st in state_in[i]:
            q = list(st)
            for ins in blocks[i]["ins"]:
                m = re.match(r"tbuffer_load_format_xyzw v\[(\d+):(\d+)\]", ins)
                if m:
                    q.append(tuple(range(int(m.group(1)), int(m.group(2)) + 1)))
                    continue
                w = re.search(r"s_waitcnt\b.*vmcnt\((\d+)\)", ins)
                if w:
                    n = int(w.group(1))
                    q = q[len(q) - n:] if n and len(q) > n else ([] if n == 0 else q)
                    continue
                pending = {r for t in q for r in t}
                hit = regs(ins) & pending
                if hit:
                    seen_bad[(i, ins)] = sorted(hit)
            outs.add(tuple(q))
        for j in succ[i]:
            new = outs - state_in[j]
            if new:
                if len(state_in[j]) > 64:
                    sys.exit("state explosion in the checker")
                state_in[j] |= new
                work.append(j)
    n_loads = sum(1 for b in blocks for ins in b["ins"] if ins.startswith("tbuffer_load_format"))
    print(f"{sym[:64]}...: {n_loads} format loads, {len(blocks)} basic blocks, "
          f"{len(seen_bad)} instructions name a register with a load in flight")
    for (i, ins), hit in list(seen_bad.items())[:12]:
        print("    block", i, blocks[i]["label"], ":", ins, hit)
    total_bad += len(seen_bad)
sys.exit(1 if total_bad else 0)
