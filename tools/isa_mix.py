"""Instruction mix per basic block of one kernel in a hipcc -save-temps .s file (CPU-side tuning aid: count VALU / MFMA / DS /
VMEM instructions of a hot loop before spending GPU time).  usage: python tools/isa_mix.py file.s <substring of kernel symbol>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = None
for mm in re.finditer(r'^(\S+):\s*(;.*)?\n', s, re.M):
    if pat in mm.group(1) and not mm.group(1).startswith('.'):
        m = mm
        break
if not m:
    raise SystemExit("kernel not found")
start = m.end()
end = s.index('.Lfunc_end', start)
body = s[start:end]
blocks, cur = [], ('entry', [])
for l in (x.strip() for x in body.split('\n')):
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur)
        cur = (l.split(':')[0], [])
    elif l and not l.startswith(';') and not l.startswith('.'):
        cur[1].append(l)
blocks.append(cur)
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter()
    for l in ins:
        i = l.split()[0]
        k = ('mfma' if i.startswith('v_mfma') else 'trans' if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)', i) else 'valu' if i.startswith('v_')
             else 'salu' if i.startswith('s_') else 'ds' if i.startswith('ds_') else 'vmem' if re.match(r'(global|buffer|flat|scratch)_', i) else 'other')
        c[k] += 1
    if len(ins) >= 8:
        print(f"{name:12s} n={len(ins):4d} {dict(c)}")
if len(sys.argv) > 3:          # dump opcode histogram of one block
    for name, ins in blocks:
        if name == sys.argv[3]:
            h = collections.Counter(l.split()[0] for l in ins)
            for k, v in h.most_common(40):
                print(f"   {k:32s} {v}")
sym = m.group(1)
for blk in re.split(r'\n  - \.agpr_count', s[s.index('amdhsa.kernels'):] if 'amdhsa.kernels' in s else ''):
    if re.search(r'\.name:\s+' + re.escape(sym) + r'\s', blk):
        for key in ('.vgpr_count', '.sgpr_count', '.vgpr_spill_count', '.group_segment_fixed_size', '.private_segment_fixed_size'):
            mm = re.search(re.escape(key) + r':\s+(\d+)', blk)
            if mm:
                print(key, mm.group(1))
