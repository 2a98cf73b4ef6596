"""Audit of the layer chain's continuous weight stream (csrc/chain.h: chain_trunk) in the COMPILED kernel: between the asm loads that
fill the run's weight queue and the counted waits that release a slot, no compiler-generated instruction may touch a queue register (a
v_mov or a scratch store of a register whose load is still in flight copies garbage - the failure mode of the first attempts, LAB_NOTES).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S climsim_amd/csrc/climsim_hip.hip -o /tmp/k.s
    python tools/trunk_audit.py /tmp/k.s _Z10k_chain_fbILi32ELb0EEv9ChainArgsS0_ [more kernel symbols]
Per run: the queue registers (destinations of the SADDR-form asm loads) and the number of non-MFMA instructions outside asm blocks that
reference them inside the run's window (must be 0).  The early primes (kernel top / the stage in front of the run) write the same
registers: round 4 also checked by hand that nothing touches them from there to the run (0 hits between line 96 and the run)."""
import re,sys
src=open(sys.argv[1]).read()
def regs_of(l):
    r=set(int(x) for x in re.findall(r'\bv(\d+)\b', l))
    for lo,hi in re.findall(r'v\[(\d+):(\d+)\]', l): r.update(range(int(lo),int(hi)+1))
    return r
for kern in sys.argv[2:]:
    m=re.search(r'^%s:.*?\.end_amdhsa_kernel' % re.escape(kern), src, re.S|re.M)
    L=m.group(0).split('\n')
    idx=[i for i,l in enumerate(L) if 'vmcnt(19)' in l]
    if not idx: print(kern,'no trunk'); continue
    clusters=[[idx[0]]]
    for i in idx[1:]:
        if i-clusters[-1][-1] > 600: clusters.append([i])
        else: clusters[-1].append(i)
    inasm=[False]*len(L); f=False
    for i,l in enumerate(L):
        if 'ASMSTART' in l: f=True
        inasm[i]=f
        if 'ASMEND' in l: f=False
    for c in clusters:
        a,b=max(0,c[0]-150),min(len(L),c[-1]+450)
        tq=set()
        for i in range(a,b):
            if inasm[i]:
                mm=re.search(r'global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[', L[i])
                if mm: tq.update(range(int(mm.group(1)), int(mm.group(2))+1))
        if not tq: continue
        # early primes: off-form asm loads into TQ registers, searching backwards from the cluster
        first=None; n_early=0
        for i in range(a,-1,-1):
            if inasm[i]:
                mm=re.search(r'global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off', L[i])
                if mm and set(range(int(mm.group(1)), int(mm.group(2))+1)) <= tq:
                    first=i; n_early+=1
                elif mm and first is not None and n_early>=16: break
            if first is not None and first-i > 400: break
        start = first if first is not None else a
        bad=0
        for i in range(start,b):
            if inasm[i]: continue
            l=L[i]
            if not l.startswith('\t') or l.strip().startswith(';'): continue
            if regs_of(l) & tq and 'v_mfma' not in l:
                bad+=1
                if bad<8: print('   !!', i+1, l)
        print(kern.split('E')[0][:28], 'trunk waits at', c[0]+1, '-', c[-1]+1, '| queue regs v%d-v%d (%d)' % (min(tq), max(tq), len(tq)),
              '| early primes: %d loads from line %s' % (n_early, first+1 if first is not None else None), '| window', start+1, '-', b, '| non-MFMA touches:', bad)
