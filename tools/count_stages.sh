#!/bin/bash
# usage: count_stages.sh <extra hipcc flags>  -- static per-stage instruction counts of the default kernel
cp ${SRC:-/root/repo/raytracer_amd/csrc/rt_photon.hip} mark.hip; sed -i "s#\"rt_device.hip.h\"#\"/root/repo/raytracer_amd/csrc/rt_device.hip.h\"#" mark.hip
python3 - <<'PY'
s=open('/tmp/isa/mark.hip').read()
s=s.replace('#include "../../include/rt_photon.h"','#include "/root/repo/include/rt_photon.h"')
def mark(s, anchor, name, after=True):
    if anchor not in s: return s
    i=s.index(anchor)
    if after: i+=len(anchor)
    return s[:i]+'\n asm volatile("; MARK_%s");\n'%name+s[i:]
s=mark(s,"        if (cls == 0u)\n        {","T_BEGIN")
s=mark(s,"        else if (cls == 1u)\n        {","E_BEGIN")
s=mark(s,"        else if (cls == 2u)\n        {","S_BEGIN")
s=mark(s,"        else if (__builtin_expect(cls == 4u, 0))\n        {","X_BEGIN")
s=mark(s,"            RT_SLAB_TALLY(tally_word)\n        }\n        else\n        {","L_BEGIN")
s=mark(s,"#undef R4\n#undef RW\n","END_STAGES")
s=mark(s,"        uint32_t age_inc;","SELECT_BEGIN",after=False)
open('/tmp/isa/mark.hip','w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -DRT_ONLY_DEFAULT_POOL "$@" -S --cuda-device-only mark.hip -o mark.s 2>&1 | grep error
awk '/^_ZN12_GLOBAL__N_120rt_trace_pool_kernelILi0ELb0ELi128ELb1ELb0EEEvNS_8PoolArgsE:/{p=1} p{print} /\.end_amdhsa_kernel/{if(p){exit}}' mark.s > main_mark.s
python3 - <<'PY'
lines=open('/tmp/isa/main_mark.s').read().splitlines()
marks=[(i,l.split('MARK_')[1]) for i,l in enumerate(lines) if 'MARK_' in l]
marks.sort()
marks.append((len(lines),'EOF'))
def count(a,b):
    v=s=ds=vm=br=rl=0
    for l in lines[a:b]:
        t=l.strip().split()
        if not t or t[0].startswith((';','.')) or t[0].endswith(':'): continue
        op=t[0]
        if op.startswith('v_'):
            v+=1
            if 'lane' in op: rl+=1
        elif op.startswith('s_'):
            s+=1
            if 'branch' in op: br+=1
        elif op.startswith('ds_'): ds+=1
        elif op.startswith(('global_','buffer_','flat_')): vm+=1
    return dict(VALU=v,SALU=s,LDS=ds,VMEM=vm,branch=br,lane=rl)
print('head', count(0,marks[0][0]))
for (a,n),(b,_) in zip(marks,marks[1:]): print(n, a, b, count(a,b))
PY
