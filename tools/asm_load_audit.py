"""Audit hipcc output for reads of registers whose asm-issued global loads may still be in flight:
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o k.s kernel.hip;  python tools/asm_load_audit.py k.s
A register written by a global_load inside an asm block stays "pending" until an asm s_waitcnt vmcnt(n) that retires it (loads
retire in order: the n youngest stay pending); any compiler-generated instruction that reads OR overwrites a pending register is reported.
The scan is BASIC-BLOCK LOCAL (the pending set is dropped at every label and branch): what it proves is that no block touches a
register between an asm load it contains and the wait that follows in the same block -- the v_mov copies hipcc may place in front of
an asm wait, an address temporary allocated over a just-requested register.  Registers that stay in flight across a loop edge
(a prefetch ring) are covered by the kernels' own wait counts, not by this screen."""
import sys,re
s=open(sys.argv[1]).read().split('\n')
# registers written by asm global_load; flag any non-asm instruction that reads them between the load and the next asm s_waitcnt
inasm=False; pending={}  # reg -> line
bad=0
def regs(tok):
    m=re.match(r'v\[(\d+):(\d+)\]',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok)
    if m: return {int(m.group(1))}
    return set()
for i,l in enumerate(s):
    t=l.strip()
    if t.startswith(';;#ASMSTART'): inasm=True; continue
    if t.startswith(';;#ASMEND'): inasm=False; continue
    if t.startswith('.LBB') or re.match(r'^_Z\w+:',t):
        pending={}                                     # a new basic block
        continue
    if not t or t.startswith((';','.')) or t.endswith(':'): continue
    parts=re.split(r'[ ,\t]+',t)
    op=parts[0]
    if op.startswith(('s_cbranch','s_branch')):
        pending={}
        continue
    if inasm and op.startswith('global_load'):
        dup=regs(parts[1]) & set(pending)
        if dup:                                        # two loads in flight into one register: hipcc thinks both results are dead
            bad+=1
            if bad<=12: print('HAZARD (dead load) line',i+1,t,'<- also loaded at line',sorted({pending[r]+1 for r in dup}))
        for r in regs(parts[1]): pending[r]=i
        continue
    if inasm and op=='s_waitcnt':
        # conservative: a wait retires the OLDEST loads only; we cannot know which -> clear regs older than N loads: approximate by clearing all
        n=int(re.search(r'vmcnt\((\d+)\)',t).group(1))
        # keep the newest n loads (by issue line, 4 regs each)
        lines=sorted(set(pending.values()))
        keep=set(lines[-n:]) if n>0 else set()
        pending={r:ln for r,ln in pending.items() if ln in keep}
        continue
    if op.startswith('s_') : continue
    # non-asm instruction: check source operands (all operands after the first for most; for stores/ds_write all)
    srcs=parts[2:] if not op.startswith(('ds_write','global_store','buffer_store')) else parts[1:]
    used=set()
    for tok in srcs: used|=regs(tok)
    hit=used & set(pending)
    if hit:
        bad+=1
        if bad<=12: print('HAZARD line',i+1,t,'<- regs loaded at line',sorted({pending[r]+1 for r in hit}))
    # a compiler-generated WRITE of a pending register is the mirror hazard: the load lands later and clobbers the new value
    if not op.startswith(('ds_write','global_store','buffer_store')) and len(parts)>1:
        wr=regs(parts[1]) & set(pending)
        if wr:
            bad+=1
            if bad<=12: print('HAZARD (overwrite) line',i+1,t,'<- regs loaded at line',sorted({pending[r]+1 for r in wr}))
            for r in wr: pending.pop(r,None)
print('hazards:',bad)
