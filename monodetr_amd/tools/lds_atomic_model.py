"""Bank model of the MSDA backward's LDS accumulation (developer tool; no GPU needed).

    python -m monodetr_amd.tools.lds_atomic_model

For one 24 x 32 level-0 core tile and each of the 8 heads at the module's initial offset pattern (head direction x point index,
+- 0.05 px of noise: tools/opbench --dist init) the own samples' `ds_add_u64` address streams of csrc/msda_fused.hip are generated
for several (cell stride, window row stride, lane layout, record order) combinations and the average number of LDS cycles per
16-lane (and 32-lane) group is counted: 64 banks x 4 bytes, a 64-bit operand takes two banks, lanes on the same bank pair in one
group serialise -- also when they hit the SAME address (atomics do not broadcast).  Result (profiles/r04_lds_atomic_model.txt):
1.8 - 1.9 for every combination -- the +-1-cell randomness of floor() puts neighbouring queries' corners on the same cell, which no
layout removes; only the rows-32-cells-apart case of heads 2 / 6 (2.56 -> 1.93) responds to padding the window rows, which is
what msda_fused.hip's row padding does.  `SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE` = 0.36 on the GPU agrees."""
import math

import numpy as np
rng=np.random.default_rng(0)
H,W=48,160; TH,TW=24,32; M=8; P=4
def samples_for_tile(m, ty, tx, noise=0.05, dist='init'):
    # own samples of level-0 queries in tile, q-major (row by row), p minor: returns top-left cell (wy,wx) relative to tile for each sample, or None if outside core
    th=m*2*math.pi/M; d=np.array([math.cos(th),math.sin(th)]); d=d/np.abs(d).max()
    out=[]
    for y in range(ty*TH,(ty+1)*TH):
        for x in range(tx*TW,(tx+1)*TW):
            for p in range(P):
                if dist=='init': off=d*(p+1)+noise*rng.standard_normal(2)
                else: off=4.0*rng.standard_normal(2)
                px=x+off[0]; py=y+off[1]
                out.append((math.floor(py)-ty*TH, math.floor(px)-tx*TW))
    return out
def conflicts(cells_list, stride_slots, tstride, lanes_per_cycle=16, layout='strided', order='qmajor', LPS=4):
    # cells_list: list of (wy,wx) for own samples in q-major order. Build groups of 16 samples; per corner c, per pr: lane addresses
    n=len(cells_list)
    idx=list(range(n))
    if order=='pmajor':
        # within each block of 64 candidates (16 queries x 4 p) reorder p-major
        idx=[]
        for b in range(0,n,64):
            blk=list(range(b,min(b+64,n)))
            idx+= sorted(blk,key=lambda i:(i%4, i//4))
    total=0; ideal=0
    for g0 in range(0,n,16):
        grp=idx[g0:g0+16]
        if len(grp)<16: break
        for c in range(4):
            dy,dx=c>>1,c&1
            for pr in range(4):
                addrs=[]
                for s in grp:
                    wy,wx=cells_list[s]; yy,xx=wy+dy,wx+dx
                    if 0<=yy<TH and 0<=xx<TW: cell=yy*tstride+xx
                    else: cell=TH*tstride+(s%8)   # sink
                    for k in range(LPS):
                        slot = cell*stride_slots + (4*k+pr if layout=='strided' else 4*pr+k)
                        addrs.append(slot)
                # process lanes in chunks
                for ch in range(0,64,lanes_per_cycle):
                    a=addrs[ch:ch+lanes_per_cycle]
                    # bank pair index = slot mod 32 ; same address -> (for atomics) serialize too
                    from collections import Counter
                    cnt=Counter([x%32 for x in a])
                    total+=max(cnt.values()); ideal+=1
    return total/ideal
for m in range(8):
    cl=samples_for_tile(m,0,1)
    r=[]
    for (stride,tst,lay,order) in [(17,32,'strided','qmajor'),(17,33,'strided','qmajor'),(17,33,'strided','pmajor'),(18,33,'contig','qmajor'),(18,33,'contig','pmajor'),(17,33,'contig','pmajor')]:
        r.append('%.2f'%conflicts(cl,stride,tst,16,lay,order))
    r2=[]
    for (stride,tst,lay,order) in [(17,32,'strided','qmajor'),(17,33,'strided','qmajor'),(17,33,'strided','pmajor'),(18,33,'contig','pmajor')]:
        r2.append('%.2f'%conflicts(cl,stride,tst,32,lay,order))
    print('head',m,'16/clk:',r,' 32/clk:',r2)
cl=samples_for_tile(0,0,1,dist='trained')
print('trained', ['%.2f'%conflicts(cl,s,t,16,l,o) for (s,t,l,o) in [(17,32,'strided','qmajor'),(17,33,'strided','qmajor'),(18,33,'contig','pmajor')]])
