"""TEST INFRASTRUCTURE (builder container only): reads the dump of subme_inner_shim.c and checks the quarter-sample step of the reference's -subme 1 refinement
against its restatement (candidate positions from the rates it was handed, SAD of the normatively interpolated samples + rate, strict '<'); findings in the shim's header.
usage: gcc -O1 -w -mgeneral-regs-only -shared -fPIC -o inner.so subme_inner_shim.c; KS265_SP_DUMP=d.bin LD_PRELOAD=./inner.so ./appencoder -i clip.yuv ... -threads 1; python subme_inner_probe.py d.bin"""
import struct, sys, numpy as np
sys.path.insert(0,'/root/repo/oracle/ref_probe')
TAPS={0:[0,0,0,64,0,0,0,0],1:[-1,4,-10,58,17,-5,1,0],2:[-1,4,-11,40,40,-11,4,-1],3:[0,1,-5,17,58,-10,4,-1]}
def interp(reg,x0,y0,W,H,qx,qy):
    ix,iy,fx,fy=qx>>2,qy>>2,qx&3,qy&3; X=x0+ix; Y=y0+iy; src=reg.astype(np.int64)
    if fx==0 and fy==0: return src[Y:Y+H,X:X+W]
    if fy==0:
        t=sum(TAPS[fx][k]*src[Y:Y+H,X-3+k:X-3+k+W] for k in range(8)); return np.clip((t+32)>>6,0,255)
    if fx==0:
        t=sum(TAPS[fy][k]*src[Y-3+k:Y-3+k+H,X:X+W] for k in range(8)); return np.clip((t+32)>>6,0,255)
    h=sum(TAPS[fx][k]*src[Y-3:Y+H+4,X-3+k:X-3+k+W] for k in range(8))-8192
    t=sum(TAPS[fy][k]*h[k:k+H,:] for k in range(8))
    return np.clip((t+8192*64+2048)>>12,0,255)
import struct, sys, numpy as np
from collections import Counter
data=open(sys.argv[1],'rb').read(); calls=[]; p=0
while p<len(data):
    hdr=struct.unpack_from('16i',data,p); p+=64
    _,idx,W,H,pux,puy,stride,mx,my,cost0,mvpx,mvpy,ox,oy,ocost,f3bc=hdr
    fe=np.frombuffer(data,np.uint8,W*H,p).reshape(H,W).astype(int); p+=W*H
    reg=np.frombuffer(data,np.uint8,(W+16)*(H+16),p).reshape(H+16,W+16); p+=(W+16)*(H+16)
    cmx=np.frombuffer(data,np.uint16,17,p).astype(int); p+=34; cmy=np.frombuffer(data,np.uint16,17,p).astype(int); p+=34
    gh=struct.unpack_from('12i',data,p); p+=48; gq=struct.unpack_from('12i',data,p); p+=48
    calls.append(dict(idx=idx,W=W,cost0=cost0,dx=ox-mx,dy=oy-my,ocost=ocost,f3bc=f3bc,fe=fe,reg=reg,cmx=cmx,cmy=cmy,gh=gh,gq=gq))
print('calls',len(calls),'hpel hook records',sum(1 for c in calls if c['gh'][0]),'qpel records',sum(1 for c in calls if c['gq'][0]))
HX=[-2,0,2,-2,2,-2,0,2]; HY=[-2,-2,-2,0,0,2,2,2]
QX=[-1,0,1,-1,1,-1,0,1]; QY=[-1,-1,-1,0,0,1,1,1]
def sad(a,b): return int(np.abs(a-b).sum())
kc=Counter(); ratematch=Counter(); costmatch=Counter(); n=0
for c in calls:
    gq=c['gq']
    if not gq[0]: continue
    k=gq[0]-1; start=gq[1]; mvc=gq[2:10]; after=gq[10]; widx=gq[11]
    kc[(k,c['W'])]+=1
    # the half-sample centre: infer from the final result if it is on the quarter grid, else from rates: try all 9 centres and see which reproduces the 8 rates in the Q order
    hi=c['gh'][11]
    cands=[(HX[hi],HY[hi])] if (c['gh'][0] and 0<=hi<8) else [(0,0)]
    found=None
    for (cx,cy) in cands:
        ok=all(0<=8+cx+QX[i]<17 and 0<=8+cy+QY[i]<17 and c['cmx'][8+cx+QX[i]]+c['cmy'][8+cy+QY[i]]==mvc[i] for i in range(8))
        if ok: found=(cx,cy); break
    ratematch[found is not None]+=1
    if found is None: continue
    cx,cy=found
    # costs of the eight candidates by SAD with real interpolation
    costs=[sad(c['fe'],interp(c['reg'],8,8,c['W'],c['W'],cx+QX[i],cy+QY[i]))+mvc[i] for i in range(8)]
    best=start; bi=-1
    for i in range(8):
        if costs[i]<best: best=costs[i]; bi=i
    costmatch[(best==after)]+=1
    if best!=after and n<8:
        n+=1; print('k',k,'centre',found,'start',start,'after',after,'widx',widx,'my best',best,bi,'costs',costs)
print('table entry used',sorted(kc.items())); print('rates reproduced in Q order around one of the 9 centres',dict(ratematch)); print('best cost after the step reproduced by SAD(real interp)+rate',dict(costmatch))
