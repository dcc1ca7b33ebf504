import re,sys
src,kname=sys.argv[1],sys.argv[2]
L=open(src).read().split('\n')
st=[i for i,l in enumerate(L) if l.startswith(kname) and l.rstrip().endswith(':') or (l.startswith(kname+':'))]
i0=[i for i,l in enumerate(L) if l.startswith(kname+':')][0]
lines=[]
for l in L[i0:]:
    lines.append(l)
    if 's_endpgm' in l: break
def cat(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_readlane') or op.startswith('v_writelane'): return 'lane'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'br'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_'): return 'vmem'
    if op.startswith('scratch_'): return 'scr'
    return 'other'
out=[]; cur=['entry',0,{}]
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m:
        out.append(cur); cur=[m.group(1),i,{}]; continue
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    op=t.split()[0]; c=cat(op)
    cur[2][c]=cur[2].get(c,0)+1
    if c=='br': cur[2].setdefault('tgt',[]).append(t.split()[-1])
out.append(cur)
print(len(lines),'lines')
for b in out:
    d=dict(b[2]); tg=d.pop('tgt',[]); tot=sum(d.values())
    if tot>=40 or d.get('mfma'): print(b[0],b[1],tot,d,tg)
