"""Build-time audit of csrc/heads_dx.hip (run by tests/test_host.py, no GPU): every inline-asm register load must be covered by a counted wait
before ANY instruction touches its destination, inside one branch-free stretch of code.

hipcc treats an asm load's destination as written at the asm statement, so it may copy / reuse the register while the load is in
flight (cdna_hip_programming.md 5.7 item 1).  The kernel is written so that cannot happen (gate registers are local to a branch-free
step); this script checks the generated ISA for it: for each run of `global_load_dwordx2 v[..], v[..], off`,
  * between the run and the first instruction that reads or writes one of its destination VGPRs there is no label and no branch, and
  * that stretch contains an `s_waitcnt vmcnt(N)` with N <= the number of vector-memory instructions issued after the run and before
    the wait (vmcnt retires in issue order: at most N outstanding then means every older operation -- the run -- has returned).
usage: python tools/audit_asm_loads.py <file.s> <kernel-name-substring>      exit code 0 = clean"""
import re, sys


def regs(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'(?<![\w\[])v(\d+)(?!\d|:)', tok):
        out.add(int(m.group(1)))
    return out


def audit(path, kernel):
    L = open(path).read().splitlines()
    s0 = next(i for i, l in enumerate(L) if kernel in l and re.match(r'^[A-Za-z_][\w$.]*:', l))      # the kernel's label line
    end = next(i for i in range(s0 + 1, len(L)) if L[i].startswith('.Lfunc_end'))
    body = [l.split(';')[0].rstrip() for l in L[s0:end]]
    is_vmem = lambda l: re.match(r'\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|flat_)', l) is not None
    gl = [i for i, l in enumerate(body) if re.match(r'\s*global_load_dwordx2 v\[', l)]
    runs, cur = [], [gl[0]]
    for a, b in zip(gl, gl[1:]):
        if all(not re.match(r'\s*(s_waitcnt vmcnt|s_barrier|v_mfma|\.LBB|s_cbranch|s_branch)', body[k]) for k in range(a, b)):
            cur.append(b)
        else:
            runs.append(cur); cur = [b]
    runs.append(cur)
    bad = 0
    for r in runs:
        dests = set()
        for i in r:
            m = re.search(r'global_load_dwordx2 v\[(\d+):(\d+)\]', body[i]); dests.update(range(int(m.group(1)), int(m.group(2)) + 1))
        vm_since, covered, jumps, touch = 0, False, [], None
        for i in range(r[-1] + 1, len(body)):
            l = body[i].strip()
            if not l or l.startswith('.') and not l.startswith('.LBB'):
                continue
            if l.startswith('.LBB') or l.startswith('s_cbranch') or l.startswith('s_branch') or l.startswith('s_setpc'):
                jumps.append((i, l))
            m = re.match(r's_waitcnt vmcnt\((\d+)\)', l)
            if m and int(m.group(1)) <= vm_since:
                covered = True
            if regs(l) & dests:
                touch = (i, l); break
            if is_vmem(l):
                vm_since += 1
        ok = covered and not jumps and touch is not None
        bad += 0 if ok else 1
        print("%s loads@%d (%d) -> v%d..v%d | first touch @%s: %-44s | %2d vmem ops between, covering wait: %s, labels / branches in the stretch: %d"
              % ("ok  " if ok else "BAD ", r[0], len(r), min(dests), max(dests), touch[0] if touch else "-", (touch[1] if touch else "-")[:44], vm_since, covered, len(jumps)))
    print("%d runs of asm register loads, %d not provably covered" % (len(runs), bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if audit(sys.argv[1], sys.argv[2]) else 0)
