import re, sys
# usage: loops.py file.s  -> innermost loops (backward branches) with scratch / readlane counts
lines=[l.rstrip('\n') for l in open(sys.argv[1])]
addr_re=re.compile(r'//\s*([0-9A-F]{12}):')
ins=[]
for l in lines:
    m=addr_re.search(l)
    if m and l.startswith('\t'):
        ins.append((int(m.group(1),16), l.strip().split('//')[0].strip()))
addr2idx={a:i for i,(a,_) in enumerate(ins)}
loops=[]
for i,(a,t) in enumerate(ins):
    m=re.match(r's_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)',t)
    if m:
        off=int(m.group(1) or m.group(2))
        if off>=32768: off-=65536
        tgt=a+4+off*4
        if tgt<=a and tgt in addr2idx:
            loops.append((addr2idx[tgt],i))
loops.sort(key=lambda x:x[1]-x[0])
for (s,e) in loops:
    body=[t for _,t in ins[s:e+1]]
    sc=sum(1 for t in body if t.startswith('scratch_'))
    rl=sum(1 for t in body if t.startswith('v_readlane') or t.startswith('v_writelane'))
    sl=sum(1 for t in body if t.startswith('s_load'))
    va=sum(1 for t in body if t.startswith('v_'))
    print(f"loop {ins[s][0]:x}-{ins[e][0]:x}: {e-s+1:5d} instr, valu {va:5d}, s_load {sl:3d}, scratch {sc:3d}, lane-spill {rl:3d}")
