#!/usr/bin/env python3
"""Static hazard check for inline-asm MFMAs in a `hipcc -S` listing (gfx950).

hipcc pads the data hazards of a `__builtin_amdgcn_mfma_*` itself, but an MFMA issued from an `asm` statement is one opaque
instruction to its hazard recognizer (cdna_hip_programming.md section 5.7 item 2): around `;;#ASMSTART` / `;;#ASMEND` it inserts at most
its fixed one-state boundary pad.  Round 4 met this as one wrong accumulator word per ~20 000 workgroups (a chain of one seed diverging;
every parity test green).  The rules below are what tools/micro/mfma_hazards.hip MEASURED on an MI355X (profiles/r05_mfma_hazards.log,
52 million lane-trials per probe, v_mfma_f32_16x16x32_f16 = 4 passes and v_mfma_f32_32x32x16_f16 = 8 passes):

  RAW-hazard  a VALU write of a register of A, B or C fewer than 2 wait states in front of the MFMA (an instruction = one wait state,
              `s_nop n` = n + 1): the MFMA reads the STALE register.  Measured: registers 0-1 of a 4-register A / B operand written by
              v_mov_b32, v_mov_b32_e64, v_mov_b64, v_xor_b32 or v_cvt_pk_f16_f32 0 or 1 state ahead (the one state being s_nop 0,
              v_nop or s_waitcnt alike): wrong in > 99 % of the trials; 2 states: never.  Registers 2-3 of the operand and all of C
              are read one state later (0 states: wrong, 1: never) — the rule keeps 2 for everything.  This is round 4's bug: hipcc's
              copies of a zeroed accumulator (v_mov_b64 into the "+v" operand) sat 0 states in front of the first asm MFMA of a block.
              (The Winograd skip tail wrote register 3 of an A operand one state ahead until round 5 — tolerated by the chip, by luck:
              profiles/r05_skip_tail_err.log shows the r04 library's results equal to the padded build's.)
  D-hazard    any instruction other than an MFMA that takes D whole as its C (the accumulate chain: hardware-interlocked) which reads or
              writes a register of D fewer than P + 4 wait states behind the MFMA (P = passes).  Measured: the LAST result register of
              the 4-pass shape is stale at 7 states and right at 8; of the 8-pass shape stale at 11, right at 12; the first register
              arrives P states earlier.
  (AB-write)  NOT a hazard, reported with --war-ab only and never part of the exit code: a VALU overwrite of A or B 0, 1, 2, 4, 8 states
              BEHIND the MFMA — behind seven independent back-to-back MFMAs (pipe full) or behind a seven-deep dependent chain alike —
              changed 0 of 52 million results: the operands are read at issue.  hipcc's own schedules do the same (294 such writes in the
              builtin-MFMA kernels of this library).  Destinations of memory loads are never counted (data returns >= 16 states later).

Wait states are followed through fall-through edges and branches in both directions.  Only v_mfma between `;;#ASMSTART` / `;;#ASMEND`
are checked (every v_mfma with --all: hipcc's own padding then shows as zero findings).  Exit code 1 if a RAW- or D-hazard is reported.
usage:
    tools/isa_hazard.py file.s [--all] [--war-ab] [--kernel SUBSTR] [--quiet]
    tools/isa_hazard.py --compile csrc/file.hip [-Dflags...]      (runs hipcc -S --offload-arch=gfx950 first)
"""
import os
import re
import subprocess
import sys
import tempfile

REG = re.compile(r'\b([va])(?:(\d+)|\[(\d+):(\d+)\])(?![\w\[])')
LABEL = re.compile(r'^([.\w$]+):')
MFMA_SHAPE = re.compile(r'v_mfma_\w+?_(\d+)x(\d+)x(\d+)_?(\w*)')

STORE_PREFIX = ('global_store', 'buffer_store', 'scratch_store', 'flat_store', 'ds_write', 'ds_store', 'exp')
LOAD_PREFIX = ('global_load', 'buffer_load', 'scratch_load', 'flat_load', 'ds_read', 'ds_load', 'ds_bpermute', 'ds_permute',
               'ds_swizzle', 's_load', 's_buffer_load')
NO_VDST = ('v_cmp', 'v_readfirstlane', 'v_readlane', 'v_nop')
WRITES_TWO = ('v_swap_b32', 'v_permlane16_swap', 'v_permlane32_swap')


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        f = m.group(1)
        if m.group(2) is not None:
            out.add((f, int(m.group(2))))
        else:
            out.update((f, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def mfma_passes(mn):
    m = MFMA_SHAPE.match(mn)
    if not m:
        return 16
    M, N, K, ty = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)
    flop = 2 * M * N * K
    if 'f64' in mn:
        return 16
    if ty.startswith(('f16', 'bf16')) or 'f16' in mn.split('_')[-1]:
        per_pass = 4096                      # 2.5 PF/s dense: 1024 FLOP per cycle and SIMD, 4 cycles per pass
    elif ty.startswith(('f8', 'bf8', 'fp8', 'i8')) or 'f8f6f4' in mn:
        per_pass = 8192
    else:
        per_pass = 256                       # f32 / xf32 operands: 64 FLOP per cycle and SIMD
    return max(1, min(16, flop // per_pass))


class Ins:
    __slots__ = ('line', 'text', 'mn', 'ops', 'writes', 'reads', 'asm', 'mfma', 'load', 'states', 'label_targets', 'terminator', 'valu')


def parse(path):
    """-> {kernel: [Ins]} plus label -> index maps"""
    kernels = {}
    cur = None
    labels = None
    in_asm = False
    for ln, raw in enumerate(open(path, errors='replace'), 1):
        line = raw.split(';;#', 1)
        if len(line) > 1:
            tag = line[1].strip()
            if tag.startswith('ASMSTART'):
                in_asm = True
            elif tag.startswith('ASMEND'):
                in_asm = False
            continue
        code = raw.split(';', 1)[0].rstrip()
        if not code.strip():
            continue
        m = LABEL.match(code)
        if m:
            name = m.group(1)
            if name.startswith('_Z') or (not name.startswith('.') and cur is None):
                cur = []
                labels = {}
                kernels[name] = (cur, labels)
            elif cur is not None:
                labels[name] = len(cur)
            continue
        if cur is None or not code.startswith(('\t', ' ')):
            continue
        t = code.strip()
        if t.startswith('.'):
            if t.startswith(('.end_amdhsa_kernel', '.section', '.amdhsa_')):
                pass
            continue
        parts = t.split(None, 1)
        mn = parts[0]
        ops = parts[1] if len(parts) > 1 else ''
        i = Ins()
        i.line, i.text, i.mn, i.ops, i.asm = ln, t, mn, ops, in_asm
        i.mfma = mn.startswith(('v_mfma', 'v_smfmac'))
        i.load = mn.startswith(LOAD_PREFIX)
        i.valu = mn.startswith('v_') and not i.mfma
        i.states = 1
        if mn == 's_nop':
            try:
                i.states = int(ops.strip(), 0) + 1
            except ValueError:
                pass
        i.terminator = mn in ('s_endpgm', 's_branch', 's_setpc_b64', 's_trap')
        i.label_targets = []
        if mn.startswith(('s_cbranch', 's_branch')):
            i.label_targets = [ops.strip()]
        first, _, rest = ops.partition(',')
        allr = regs_of(ops)
        if mn.startswith('s_') or mn.startswith(STORE_PREFIX) or mn.startswith(NO_VDST):
            i.writes, i.reads = set(), allr
        elif '_atomic' in mn:
            ret = bool(re.search(r'\b(sc0|glc)\b', ops))
            i.writes = regs_of(first) if ret else set()
            i.reads = regs_of(rest) if ret else allr
        elif i.load:
            lds_form = bool(re.search(r'\blds\b', ops)) or '_lds_' in mn
            i.writes = set() if lds_form else regs_of(first)
            i.reads = allr if lds_form else regs_of(rest)
        elif mn.startswith(WRITES_TWO):
            second = rest.split(',', 1)[0]
            i.writes = regs_of(first) | regs_of(second)
            i.reads = allr
        else:
            i.writes = regs_of(first)
            i.reads = regs_of(rest)
            if i.mfma or 'mac' in mn or 'dpp' in t or 'row_' in t or 'quad_perm' in t or mn.startswith(('v_dot', 'v_pk_fmac')):
                i.reads = i.reads | (i.writes if not i.mfma else set())
        cur.append(i)
    return kernels


def mfma_operands(i):
    ops = [o.strip() for o in i.ops.split(',')]
    d = regs_of(ops[0])
    a = regs_of(ops[1]) if len(ops) > 1 else set()
    b = regs_of(ops[2]) if len(ops) > 2 else set()
    c = regs_of(ops[3]) if len(ops) > 3 else set()
    return d, a, b, c


def fmt(rs):
    out = []
    for f in 'va':
        xs = sorted(n for (g, n) in rs if g == f)
        k = 0
        while k < len(xs):
            j = k
            while j + 1 < len(xs) and xs[j + 1] == xs[j] + 1:
                j += 1
            out.append(f'{f}{xs[k]}' if j == k else f'{f}[{xs[k]}:{xs[j]}]')
            k = j + 1
    return ','.join(out)


def check_kernel(name, ins, labels, all_mfma=False, raw_states=2, war_ab=False):
    findings = []
    n = len(ins)
    # predecessors: the instruction above (unless it ends its block for good) and every branch that names this label
    preds = {}
    for j, x in enumerate(ins):
        if j + 1 < n and not x.terminator:
            preds.setdefault(j + 1, []).append(j)
        for t in x.label_targets:
            if t in labels and labels[t] < n:
                preds.setdefault(labels[t], []).append(j)
    for idx, m in enumerate(ins):
        if not m.mfma or not (m.asm or all_mfma):
            continue
        D, A, B, C = mfma_operands(m)
        P = mfma_passes(m.mn)
        W = P + 4
        # ---- behind the MFMA
        seen = {}
        work = [] if m.terminator else [(idx + 1, 0)]
        while work:
            j, used = work.pop()
            while j < n and used < W:
                if seen.get(j, 1 << 30) <= used:
                    break
                seen[j] = used
                x = ins[j]
                if x.mfma:
                    xd, xa, xb, xc = mfma_operands(x)
                    # an MFMA that takes D whole as its C needs no pad (the accumulate chain, whatever it writes); one that reads D as A / B,
                    # reads part of it as C, or writes over it without reading it does
                    touch = ((xa | xb) & D) | ((xc & D) if xc != D else set()) | ((xd & D) if xc != D else set())
                    if touch:
                        findings.append(('D-hazard', m, x, used, touch))
                else:
                    touch = (x.writes | x.reads) & D
                    if touch:
                        findings.append(('D-hazard', m, x, used, touch))
                    if war_ab and not x.load:
                        w = x.writes & (A | B)
                        if w:
                            findings.append(('AB-write', m, x, used, w))
                used += x.states
                for t in x.label_targets:
                    if t in labels:
                        work.append((labels[t], used))
                if x.terminator:
                    break
                j += 1
        # ---- in front of it
        seen = {}
        work = [(j, 0) for j in preds.get(idx, [])]
        while work:
            j, used = work.pop()
            if used >= raw_states or seen.get(j, 1 << 30) <= used:
                continue
            seen[j] = used
            x = ins[j]
            if x.valu:
                w = x.writes & (A | B | C)
                if w:
                    findings.append(('RAW-hazard', m, x, used, w))
            for k in preds.get(j, []):
                work.append((k, used + x.states))
    # one line per (kind, mfma line, other line)
    uniq = {}
    for f in findings:
        uniq.setdefault((f[0], f[1].line, f[2].line), f)
    return list(uniq.values())


def compile_to_s(src, flags):
    here = os.path.dirname(os.path.abspath(__file__))
    inc = os.path.join(here, '..', 'include')
    out = tempfile.NamedTemporaryFile(suffix='.s', delete=False).name
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', f'-I{inc}', f'-I{os.path.dirname(os.path.abspath(src))}',
           '-Wno-unused-function', '-ffp-contract=off', '-S', '--cuda-device-only', *flags, src, '-o', out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def run(path, all_mfma=False, kernel=None, quiet=False, out=sys.stdout, war_ab=False):
    kernels = parse(path)
    total = 0
    sites = 0
    for name, (ins, labels) in kernels.items():
        if kernel and kernel not in name:
            continue
        n_m = sum(1 for i in ins if i.mfma and (i.asm or all_mfma))
        if not n_m:
            continue
        sites += n_m
        fs = check_kernel(name, ins, labels, all_mfma, war_ab=war_ab)
        total += sum(1 for f in fs if f[0] != 'AB-write')
        if not quiet:
            print(f'{name}: {n_m} {"" if all_mfma else "inline-asm "}MFMAs, {len(fs)} findings', file=out)
            for kind, m, x, used, regs in sorted(fs, key=lambda f: (f[1].line, f[2].line))[:40]:
                where = 'behind' if kind != 'RAW-hazard' else 'in front of'
                print(f'   {kind:10s} line {x.line}: `{x.text}` touches {fmt(regs)} {used} wait state(s) {where} line {m.line}: `{m.text}`', file=out)
            if len(fs) > 40:
                print(f'   ... {len(fs) - 40} more', file=out)
    if not quiet:
        print(f'{os.path.basename(path)}: {sites} MFMA sites checked, {total} hazards', file=out)
    return sites, total


def main(argv):
    args = [a for a in argv if not a.startswith('--')]
    flags = [a for a in argv if a.startswith('--')]
    kernel = None
    if '--kernel' in argv:
        kernel = argv[argv.index('--kernel') + 1]
        args.remove(kernel)
    if '--compile' in flags:
        src = args[0]
        path = compile_to_s(src, [a for a in args[1:]] + [a for a in flags if a.startswith('--D')])
    else:
        path = args[0]
    sites, total = run(path, all_mfma='--all' in flags, kernel=kernel, quiet='--quiet' in flags, war_ab='--war-ab' in flags)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
