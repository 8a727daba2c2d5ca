"""Static check of the device assembly, run by `make` on every build: no instruction may touch a
register that an in-flight INLINE-ASM load still owes.

Why ("The inline-asm load hazard"): to the compiler the registers an inline-asm load writes are
defined at the asm statement.  Between the load and the wait that covers it hipcc is free to copy
them (it then keeps STALE data: a silently wrong query fragment), to spill them, or to hand the
registers to something else (the load then lands in somebody's address pair: the GPU memory faults
of rounds 2 and 3; round 4 met the copy form - `v_accvgpr_write_b32 a84, v104` right behind
`global_load_dwordx4 v[104:107]` - in a build whose register pressure had changed).  The source
cannot rule that out; the emitted code can be checked, and a finding fails the build.

What is checked, per kernel that contains an asm `global_load_*`:
  * the kernel is cut into basic blocks (labels, s_branch / s_cbranch_*, s_endpgm); a data-flow pass
    over its control-flow graph runs to a fixed point, loops included (no line window);
  * state = the registers owed by asm loads, each with the SMALLEST number of vmcnt-counted
    operations issued behind its load on any path (loads return in order: `s_waitcnt vmcnt(N)`
    covers a load iff N <= that number);
  * ANY instruction that reads or writes an owed register is a finding: copies (v_mov, v_accvgpr_write),
    spills (scratch_store), address use, recycling; so is an asm load into a register that is owed;
  * the run-time wait selector (`wait_vm_dyn`, dma_scan.hpp) is a switch over `s_waitcnt vmcnt(K)`
    cases that a static pass cannot follow: it is bracketed by `; cslam.wait_dyn.begin / .end`
    comments.  A plain selector covers nothing as far as this check is concerned (every case is
    applied, the join keeps what any path keeps); a selector whose end marker says
    `covers-asm-loads` is the kernel's own statement that its count covers every asm load issued
    before it - taken at its word (a wrong count shows up as wrong bits in the parity tests, and does
    not depend on the compiler);
  * indirect control flow (s_setpc / s_swappc) in such a kernel is a finding: the graph would be unknown.
Also reported (not findings): scratch traffic of kernels with asm loads - a scratch reload is a
`vmcnt(0)` in the middle of a counted ring.

usage: check_asm_load_hazard.py file.s [file.s ...] [--kernels SUBSTR] [--quiet]   exit 1 on findings
"""
import re
import sys

VMEM_RE = re.compile(r"^(global|buffer|scratch|flat|tbuffer|image)_")
REG_RE = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
BRANCH_RE = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(\S+)")


def regs(text):
    out = set()
    for m in REG_RE.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


class Ins:
    __slots__ = ("ln", "text", "op", "ops", "in_asm", "marker")

    def __init__(self, ln, text, in_asm, marker=None):
        self.ln, self.text, self.in_asm, self.marker = ln, text, in_asm, marker
        parts = text.split(None, 1)
        self.op = parts[0] if parts else ""
        self.ops = parts[1] if len(parts) > 1 else ""


def parse_kernels(path):
    """{kernel name: [items]} where an item is ('label', name) or an Ins."""
    kernels, cur, name, in_asm = {}, None, None, False
    for ln, raw in enumerate(open(path), 1):
        s = raw.strip()
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", s)
        if m and not s.startswith(".L"):
            name, cur, in_asm = m.group(1), [], False
            kernels[name] = cur
            continue
        if cur is None:
            continue
        if s.startswith(".Lfunc_end"):
            cur, name = None, None
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if in_asm and s.startswith("; cslam.wait_dyn."):
            cur.append(Ins(ln, s, True, marker=s[2:].strip()))
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            cur.append(("label", m.group(1)))
            continue
        body = s.split(";")[0].strip()
        if not body or body.startswith("."):
            continue
        cur.append(Ins(ln, body, in_asm))
    return kernels


def build_blocks(items):
    """basic blocks: list of dict(ins=[...], succ=[block indices]); returns (blocks, problems)."""
    blocks, label_at, cur = [], {}, None

    def new_block():
        nonlocal cur
        cur = {"ins": [], "succ": [], "fall": True}
        blocks.append(cur)

    new_block()
    for it in items:
        if isinstance(it, tuple):
            if cur["ins"]:
                new_block()
            label_at[it[1]] = len(blocks) - 1
            continue
        cur["ins"].append(it)
        if BRANCH_RE.match(it.text) or it.op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_call_b64"):
            new_block()
    problems = []
    for i, b in enumerate(blocks):
        last = b["ins"][-1] if b["ins"] else None
        nxt = i + 1 if i + 1 < len(blocks) else None
        if last is None:
            if nxt is not None:
                b["succ"].append(nxt)
            continue
        m = BRANCH_RE.match(last.text)
        if m:
            tgt = label_at.get(m.group(2))
            if tgt is None:
                problems.append((last.ln, last.text, "branch target not found"))
            else:
                b["succ"].append(tgt)
            if m.group(1) != "s_branch" and nxt is not None:
                b["succ"].append(nxt)
        elif last.op == "s_endpgm":
            pass
        elif last.op in ("s_setpc_b64", "s_swappc_b64", "s_call_b64"):
            problems.append((last.ln, last.text, "indirect control flow in a kernel with inline-asm loads"))
        elif nxt is not None:
            b["succ"].append(nxt)
    return blocks, problems


def transfer(state, ins, findings, kernel, record):
    """state: {reg: (younger, load line)}; returns the state behind `ins`."""
    if ins.marker is not None:
        if ins.marker.startswith("cslam.wait_dyn.end") and "covers-asm-loads" in ins.marker:
            return {}
        return state
    m = re.search(r"vmcnt\((\d+)\)", ins.text) if ins.op == "s_waitcnt" else None
    if ins.op == "s_waitcnt":
        if m:
            n = int(m.group(1))
            return {r: v for r, v in state.items() if v[0] < n}
        return state
    is_vmem = bool(VMEM_RE.match(ins.op))
    is_asm_load = ins.in_asm and ins.op.startswith("global_load_") and "_lds" not in ins.op
    touched = regs(ins.ops)
    hit = touched & set(state)
    if hit and record:
        r0 = min(hit)
        findings.append((kernel, ins.ln, ins.text, sorted(f"{a}{i}" for a, i in hit), state[r0][1]))
    if is_vmem:
        state = {r: (y + 1, ln) for r, (y, ln) in state.items()}
    if is_asm_load:
        state = dict(state)
        for r in regs(ins.ops.split(",")[0]):
            state[r] = (0, ins.ln)
    return state


def join(a, b):
    """a |= b with the smaller younger-count; returns True when a changed."""
    changed = False
    for r, v in b.items():
        if r not in a or v[0] < a[r][0]:
            a[r] = v
            changed = True
    return changed


def check_kernel(name, items):
    if not any(isinstance(i, Ins) and i.in_asm and i.op.startswith("global_load_") and "_lds" not in i.op for i in items):
        return None
    blocks, problems = build_blocks(items)
    findings = [(name, ln, text, [], why) for ln, text, why in problems]
    entry = [dict() for _ in blocks]
    seen = [False] * len(blocks)
    seen[0] = True
    work = [0]
    while work:
        i = work.pop()
        st = dict(entry[i])
        for ins in blocks[i]["ins"]:
            st = transfer(st, ins, findings, name, False)
        for s in blocks[i]["succ"]:
            if join(entry[s], st) or not seen[s]:
                seen[s] = True
                if s not in work:
                    work.append(s)
    for i, b in enumerate(blocks):      # fixed point reached: report
        if not seen[i]:
            continue
        st = dict(entry[i])
        for ins in b["ins"]:
            st = transfer(st, ins, findings, name, True)
    scratch = sum(1 for it in items if isinstance(it, Ins) and it.op.startswith("scratch_"))
    nloads = sum(1 for it in items if isinstance(it, Ins) and it.in_asm and it.op.startswith("global_load_") and "_lds" not in it.op)
    return findings, scratch, nloads


def check(path, want=None, verbose=True):
    """All findings of the kernels of `path` that contain inline-asm loads (`want`: name substring)."""
    out = []
    checked = 0
    for name, items in parse_kernels(path).items():
        if want and want not in name:
            continue
        res = check_kernel(name, items)
        if res is None:
            continue
        checked += 1
        findings, scratch, nloads = res
        out.extend(findings)
        if verbose and scratch:
            print(f"note: {name[:80]}: {scratch} scratch operations beside {nloads} asm loads")
    if verbose:
        for k, ln, ins, t, src in out[:16]:
            why = f"touches {','.join(t)} still owed by the asm load of line {src}" if t else str(src)
            print(f"{path}:{ln}: {k[:60]}..: `{ins[:70]}` {why}")
        print(f"{path}: {checked} kernel(s) with inline-asm loads, {len(out)} finding(s)")
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    want = None
    if "--kernels" in sys.argv:
        want = sys.argv[sys.argv.index("--kernels") + 1]
        args = [a for a in args if a != want]
    quiet = "--quiet" in sys.argv
    total = 0
    for f in args:
        total += len(check(f, want, verbose=not quiet))
    if not args:
        print(__doc__)
        sys.exit(2)
    sys.exit(1 if total else 0)
