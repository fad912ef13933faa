"""Static instruction mix of /tmp/kernel.s (tools/disasm.sh), split at the barriers:  python3 tools/isa_mix.py"""
from collections import Counter
lines = [l.strip() for l in open('/tmp/kernel.s') if l.strip()]
seg = Counter(); segs = []; cur = 'start'
for l in lines:
    op = l.split()[0]
    if op.endswith(':'): continue
    if op == 's_barrier':
        segs.append((cur, dict(seg))); seg = Counter(); cur = 'after barrier %d' % len(segs); continue
    if op.startswith('s_load'): k = 'smem'
    elif op.startswith('s_waitcnt') or op.startswith('s_nop'): k = 'wait'
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): k = 'branch'
    elif op.startswith('s_'): k = 'salu'
    elif op.startswith('v_'): k = 'valu'
    elif op.startswith('ds_'): k = 'lds'
    else: k = 'vmem'
    seg[k] += 1
segs.append((cur, dict(seg)))
for s in segs: print(s)
