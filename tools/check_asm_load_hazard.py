"""Static check of the device assembly for the hazard that cost two rounds a GPU fault each
(DESIGN.md section 5, item 10).  The registers an INLINE-ASM load writes are, to the compiler, defined
at the asm statement; between the load and the wait that covers it hipcc is free to copy them or
to hand them to something else.  Round 3 caught it in the act: a variant of k_scan_dma that issued
LDS-DMA pieces before waiting for its asm query loads had `global_load_dwordx4 v[6:9]` in flight
while `v[6:7]` / `v[8:9]` were recycled as the DMA's address pair - the query data landed in the
address registers and the next piece went to a wild address (HSA_STATUS_ERROR_MEMORY_APERTURE_
VIOLATION, the fault of round 2's branch-free issue variants as well).

Rule checked per kernel, in text order (fall-through order; the vmcnt counter is simulated, a wait
right behind a conditional branch is taken as skippable, the run-time wait selector - a run of
`s_waitcnt vmcnt(N)` cases - is trusted to cover the loads it follows): while an asm `global_load_*` is not yet
covered by a wait, no other instruction may touch its destination registers.
usage: check_asm_load_hazard.py file.s [kernel-name-substring]    exit status 1 on findings"""
import re
import sys


WINDOW = 450   # lines


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def check(path, want="k_scan", verbose=True):
    src = open(path).read().split("\n")
    findings = []
    kernel, in_asm, pending, seq = None, False, {}, 0
    prev_cbranch, wait_group = False, []   # was the previous instruction a conditional branch; pending run of waits

    def flush_waits():
        nonlocal wait_group, pending
        if wait_group:
            if len(wait_group) >= 10:
                # the run-time selector (wait_vm_dyn): its count is the kernel's own bookkeeping of
                # what was issued behind the loads it is there to cover - taken as covering them
                pending = {}
            else:
                done = seq - max(wait_group)
                pending = {r: (s, ln) for r, (s, ln) in pending.items() if s > done}
            wait_group = []

    for ln, line in enumerate(src, 1):
        s = line.strip()
        km = re.match(r"(_Z\w+):", s)
        if km:
            flush_waits()
            kernel, pending, seq = km.group(1), {}, 0
        if kernel is None or want not in kernel:
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or (s.startswith(".") and not s.startswith(".LBB")):
            continue
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", s)
        if m:
            # a lone wait right behind a conditional branch may be jumped over; the cases of the run-time
            # selector sit behind branches too, but each is followed by a branch to the selector's end
            nxt = next((x.strip() for x in src[ln:ln + 4] if x.strip() and not x.strip().startswith(";")), "")
            if not prev_cbranch or nxt.startswith("s_branch") or nxt.startswith("s_waitcnt"):
                wait_group.append(int(m.group(1)))
            prev_cbranch = False
            continue
        if s.startswith("s_branch") and wait_group:
            continue                             # inside a selector
        if s.startswith(".LBB"):
            continue
        prev_cbranch = s.startswith("s_cbranch")
        if s.startswith("s_"):
            continue                             # scalar code inside a selector does not end it
        flush_waits()
        body = s.split(";")[0]
        is_vmem = bool(re.match(r"(global|buffer|scratch|flat)_(load|store|atomic)", body))
        if is_vmem:
            seq += 1
        if in_asm and body.startswith("global_load_dword"):
            for r in regs(body.split(",")[0]):
                pending[r] = (seq, ln)
            continue
        # the failure mode is RECYCLING: a compiler-generated instruction that WRITES a register the load
        # still owes (first operand of anything but a store), or a memory instruction that uses one as
        # its address.  Plain reads are what the consumers do behind the kernel's own counted waits.
        parts = body.split(None, 1)
        ops = parts[1] if len(parts) > 1 else ""
        first = ops.split(",")[0]
        if re.match(r"(global|buffer|scratch|flat)_", body):
            touched = regs(ops if "_load_lds" in body or "_store" in body else ",".join(ops.split(",")[1:])) & set(pending)
        else:
            touched = regs(first) & set(pending)
        # text order stops being execution order across the kernel's loops: an entry is only held for
        # the few hundred lines in which its covering wait must appear
        touched = {r for r in touched if ln - pending[r][1] <= WINDOW}
        if touched:
            r0 = min(touched)
            findings.append((kernel, ln, body.strip(), sorted(touched), pending[r0][1]))
    if verbose:
        for k, ln, ins, t, src_ln in findings[:12]:
            print(f"{k[:70]}.. line {ln}: `{ins[:72]}` touches v{t} still owed by the asm load of line {src_ln}")
        print(f"{path}: {len(findings)} finding(s) in kernels matching '{want}'")
    return findings


if __name__ == "__main__":
    f = check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_scan")
    sys.exit(1 if f else 0)
