"""Audit of the layer chain's continuous weight stream (csrc/chain.h: chain_trunk) in the COMPILED kernel: between the asm loads that
fill the weight queue and the counted waits that release a slot, no compiler-generated instruction may touch a queue register (a v_mov of
a register whose load is still in flight copies garbage - the failure mode of the first attempts, see LAB_NOTES).  Usage:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S climsim_amd/csrc/climsim_hip.hip -o /tmp/k.s
    python tools/trunk_audit.py /tmp/k.s _Z10k_chain_fbILi32ELb0EEv9ChainArgsS0_ [more kernel symbols]
Prints, per trunk region, the queue registers and the number of non-MFMA instructions outside asm blocks that reference them (must be 0)."""
import re,sys
src=open(sys.argv[1]).read()
for kern in sys.argv[2:]:
    m=re.search(r'^%s:.*?\.end_amdhsa_kernel' % re.escape(kern), src, re.S|re.M)
    L=m.group(0).split('\n')
    idx=[i for i,l in enumerate(L) if 'vmcnt(19)' in l]
    if not idx: print(kern,'no trunk'); continue
    # split into clusters (fwd / bwd)
    clusters=[[idx[0]]]
    for i in idx[1:]:
        if i-clusters[-1][-1] > 600: clusters.append([i])
        else: clusters[-1].append(i)
    for c in clusters:
        a,b=c[0]-120,c[-1]+450
        # the region ends at the stage loop's BACK EDGE (a branch to a label defined in front of the cluster): behind the loop the run
        # is over, the queue is empty and its registers are anybody's (round 5: the +450 window reached into the code behind the loop)
        labels={L[i].split(':')[0].strip():i for i in range(max(0,a-4000),c[0]) if re.match(r'^\.LBB\w+:', L[i])}
        for i in range(c[-1],min(len(L),c[-1]+450)):
            mm=re.match(r'\ts_branch (\.LBB\w+)', L[i])
            if mm and mm.group(1) in labels: b=i+1; break
        dest=set(); inasm=False
        for i in range(a,b):
            if 'ASMSTART' in L[i]: inasm=True
            elif 'ASMEND' in L[i]: inasm=False
            elif inasm:
                mm=re.search(r'global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[', L[i])
                if mm: dest.update(range(int(mm.group(1)), int(mm.group(2))+1))
        bad=0; inasm=False
        for i in range(a,b):
            if 'ASMSTART' in L[i]: inasm=True; continue
            if 'ASMEND' in L[i]: inasm=False; continue
            if inasm: continue
            l=L[i]
            if not l.startswith('\t') or l.strip().startswith(';'): continue
            regs=set(int(x) for x in re.findall(r'\bv(\d+)\b', l))
            for lo,hi in re.findall(r'v\[(\d+):(\d+)\]', l): regs.update(range(int(lo),int(hi)+1))
            if regs & dest and 'v_mfma' not in l:
                bad+=1
                if bad<6: print('   !!', i+1, l)
        print(kern, 'cluster lines', c[0]+1, '-', c[-1]+1, 'queue regs', min(dest), '-', max(dest), len(dest), 'non-MFMA touches:', bad)
