"""Static check on the compiled device assembly of the kernels whose LOADER waves keep global loads in flight across
hand-counted `s_waitcnt vmcnt(N)` (inline-assembly `buffer_load_dwordx4` into registers): no instruction may READ such a
register -- or WRITE anything else to it -- before the wait that covers its load.  The compiler does not know the load is asynchronous and is free to COPY
the register (live-range splits, tied inline-assembly operands) -- a copy of stale data, wrong only when memory is slow.

The loader region (from its `s_setprio` to the next `s_endpgm`) is scanned in text order with the in-order queue of
vector-memory requests simulated: `s_waitcnt vmcnt(N)` completes all but the newest N.  Innermost loops are scanned twice,
so that a read at the top of a trip sees the loads issued at the bottom of the previous one.

    python scripts/check_inflight_regs.py <file.s> <kernel substring>      (exit code 1 + a listing on a violation)"""
import re, sys


def regs_of(txt):
    out = set()
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', txt):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r'(?<![\[:\w])v(\d+)\b', txt):
        out.add(int(a))
    return out


def loader_region(body):
    for i, l in enumerate(body):
        if 's_setprio' in l:
            for j in range(i, len(body)):
                if 's_endpgm' in body[j] and not any('s_barrier' in x for x in body[j:j + 400] if 's_endpgm' not in x):
                    return i, j + 1
            return i, len(body)
    return None


def unroll_loops(lines):
    """duplicate every innermost loop body once (label '... Inner Loop Header' .. last branch back to it)"""
    out, i = [], 0
    while i < len(lines):
        m = re.match(r'^(\.LBB\d+_\d+):.*Loop Header', lines[i][1])
        if m:
            lab = m.group(1)
            last = None
            for j in range(i + 1, len(lines)):
                if re.search(r'\bs_c?branch\w*\s+%s\b' % re.escape(lab), lines[j][1]):
                    last = j
            if last is not None:
                seg = lines[i:last + 1]
                out += seg + seg
                i = last + 1
                continue
        out.append(lines[i]); i += 1
    return out


def check(path, kernel, collect=None):
    s = open(path).read()
    total = 0
    for m in re.finditer(r'^(_Z[^\n]*%s[^:\n]*):' % re.escape(kernel), s, re.M):
        start = m.start(); end = s.index('.Lfunc_end', start)
        body = s[start:end].split('\n')
        reg = loader_region(body)
        if reg is None:
            print('%s: no loader region' % m.group(1)[:70]); continue
        lines = unroll_loops([(i, body[i].strip()) for i in range(reg[0], reg[1])])
        queue, bad = [], []   # queue of sets of destination registers (empty set: LDS-DMA / store)
        for i, t in lines:
            if not t or t.startswith((';', '.')):
                continue
            if t.startswith('s_waitcnt') and 'vmcnt' in t:
                n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
                queue = queue[len(queue) - n:] if n else []
                continue
            if t.startswith(('buffer_load', 'global_load')):
                mm = re.match(r'\S+ (v\[\d+:\d+\]|v\d+),', t)
                queue.append(regs_of(mm.group(1)) if (mm and ' lds' not in t) else set())
                continue
            if t.startswith(('buffer_store', 'global_store')):
                queue.append(set()); continue
            parts = t.split(None, 1)
            if len(parts) < 2 or not parts[0].startswith(('v_', 'ds_', 'scratch_')):
                continue
            ops = parts[1].split(',')
            is_store = parts[0].startswith(('ds_write', 'scratch_store'))
            src = regs_of(parts[1]) if is_store else regs_of(','.join(ops[1:]))
            inflight = set().union(*queue) if queue else set()
            if src & inflight:
                bad.append((i, t))
            # ... nor may anything else be WRITTEN to such a register (the compiler handed it to another value -- after a
            # copy, or because the loaded value is never used --: the load lands later and overwrites that value)
            dst = set() if is_store else regs_of(ops[0])
            if dst & inflight:
                bad.append((i, t + '    ; WRITE to a register with a load in flight'))
        seen, uniq = set(), []
        for b in bad:
            if b not in seen:
                seen.add(b); uniq.append(b)
        print('%s: %d reads of registers whose load is still in flight' % (m.group(1)[:70], len(uniq)))
        for i, t in uniq[:16]:
            print('    line %d: %s' % (i, t))
        total += len(uniq)
        if collect is not None:
            collect.extend(t for _, t in uniq)
    return total


def check_dma_addr(path, kernel):
    """A precaution kept from the bring-up of csrc/convtr_s3.hpp, NOT a hardware requirement: in the split-bf16 kernels the
    address register of a `buffer_load_dwordx4 ... lds` copy is not written while the copy may be in flight.  (The corrupted
    copies that prompted it came from a register handed out under an in-flight load -- rule one above --;
    scripts/micro/lds_dma_hazards.hip shows that the copies themselves tolerate the rewrite.)  In the loader region, a vector
    instruction that writes the address register of a copy issued since the last `s_waitcnt vmcnt(0)` is reported."""
    s = open(path).read()
    total = 0
    for m in re.finditer(r'^(_Z[^\n]*%s[^:\n]*):' % re.escape(kernel), s, re.M):
        start = m.start(); end = s.index('.Lfunc_end', start)
        body = s[start:end].split('\n')
        reg = loader_region(body)
        if reg is None:
            print('%s: no loader region' % m.group(1)[:70]); continue
        lines = unroll_loops([(i, body[i].strip()) for i in range(reg[0], reg[1])])
        inflight, bad = set(), []
        for i, t in lines:
            if not t or t.startswith((';', '.')):
                continue
            if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
                inflight = set(); continue
            mm = re.match(r'buffer_load_dword\w* (v\d+), s\[.*\blds\b', t)
            if mm:
                inflight |= regs_of(mm.group(1)); continue
            parts = t.split(None, 1)
            if len(parts) < 2 or not parts[0].startswith(('v_', 'ds_read', 'buffer_load', 'global_load', 'scratch_load')):
                continue
            dst = regs_of(parts[1].split(',')[0])
            if dst & inflight:
                bad.append((i, t))
        uniq = sorted(set(bad))
        print('%s: %d writes to the address register of an LDS-DMA copy in flight' % (m.group(1)[:70], len(uniq)))
        for i, t in uniq[:16]:
            print('    line %d: %s' % (i, t))
        total += len(uniq)
    return total


if __name__ == '__main__':
    bad = check(sys.argv[1], sys.argv[2]) + check_dma_addr(sys.argv[1], sys.argv[2])
    sys.exit(1 if bad else 0)
