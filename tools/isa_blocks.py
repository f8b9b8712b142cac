"""Per basic block of every kernel in a hipcc -S listing: MFMAs, scratch (spill) traffic, waits.  usage: python tools/isa_blocks.py file.s [min_mfma]
A block with MFMAs AND scratch loads is what the single-wave conv kernels must not have (a scratch reload is followed by vmcnt(0))."""
import re, sys
path = sys.argv[1]
fn = None
blocks = []
cur = None
for ln, line in enumerate(open(path), 1):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        fn = m.group(1); cur = [fn, 'entry', ln, 0, 0, 0, 0]; blocks.append(cur); continue
    m = re.match(r'^(\.LBB\d+_\d+):', line)
    if m and fn:
        cur = [fn, m.group(1), ln, 0, 0, 0, 0]; blocks.append(cur); continue
    if cur is None: continue
    t = line.strip()
    if t.startswith('v_mfma'): cur[3] += 1
    elif t.startswith('scratch_load'): cur[4] += 1
    elif t.startswith('scratch_store'): cur[5] += 1
    elif t.startswith('s_waitcnt') and 'vmcnt(0)' in t: cur[6] += 1
last = None
for b in blocks:
    if b[0] != last:
        print(b[0]); last = b[0]
    if b[3] or b[4] or b[5]:
        print(f"   {b[1]:12s} line {b[2]:7d}  mfma {b[3]:5d}  scratch_load {b[4]:4d}  scratch_store {b[5]:4d}  vmcnt(0) {b[6]:3d}")
