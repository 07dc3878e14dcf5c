"""TEST INFRASTRUCTURE (builder container only): replay the sub-pel refinement calls recorded by subme_trace_shim.c under several hypotheses about the
reference's measure (Hadamard or SAD, baseline = the start cost or the recomputed centre) and print how many calls reproduce its vector.
usage: gcc -O1 -w -shared -fPIC -o subme.so subme_trace_shim.c; KS265_SP_DUMP=dump.bin LD_PRELOAD=./subme.so ./appencoder -i clip.yuv ... -threads 1; python subme_replay.py dump.bin"""
import struct, sys, ctypes as C, numpy as np, itertools
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle_lib import lib, ptr
o=lib()
HX=[int(x) for x in np.frombuffer(open('/root/reference/ubuntu_x64/appencoder','rb').read()[0xe61f1:0xe61f9],np.int8)]
HY=[int(x) for x in np.frombuffer(open('/root/reference/ubuntu_x64/appencoder','rb').read()[0xe61e9:0xe61f1],np.int8)]
print("hpel_x",HX,"hpel_y",HY)
TAPS={0:[0,0,0,64,0,0,0,0],1:[-1,4,-10,58,17,-5,1,0],2:[-1,4,-11,40,40,-11,4,-1],3:[0,1,-5,17,58,-10,4,-1]}
def interp(reg, x0, y0, W, H, qx, qy):
    # reg: region array, (x0,y0) = position of the block's integer origin inside reg; returns W x H prediction at quarter offset (qx,qy) relative to that origin
    ix, iy, fx, fy = qx>>2, qy>>2, qx&3, qy&3
    X=x0+ix; Y=y0+iy
    src=reg.astype(np.int64)
    if fx==0 and fy==0: return src[Y:Y+H, X:X+W]
    if fy==0:
        t=sum(TAPS[fx][k]*src[Y:Y+H, X-3+k:X-3+k+W] for k in range(8)); return np.clip((t+32)>>6,0,255)
    if fx==0:
        t=sum(TAPS[fy][k]*src[Y-3+k:Y-3+k+H, X:X+W] for k in range(8)); return np.clip((t+32)>>6,0,255)
    h=sum(TAPS[fx][k]*src[Y-3:Y+H+4, X-3+k:X-3+k+W] for k in range(8)) - 8192   # 16-bit intermediate with offset as HM: (sum - 8192) >> 0 for 8-bit
    t=sum(TAPS[fy][k]*h[k:k+H,:] for k in range(8))
    return np.clip((t + 8192*64 + 2048)>>12,0,255)
def had(a,b,n):
    A=np.ascontiguousarray(a.astype(np.uint8)); B=np.ascontiguousarray(b.astype(np.uint8))
    return int(o.ks265o_had(ptr(A),ptr(B),C.c_long(n) if False else n, n, n, n))
o.ks265o_had.restype=C.c_uint32

data=open(sys.argv[1],'rb').read()
calls=[]; p=0
while p < len(data):
    hdr=struct.unpack_from('16i',data,p); p+=64
    _,idx,W,H,pux,puy,stride,mx,my,cost0,mvpx,mvpy,ox,oy,ocost,f3bc=hdr
    fe=np.frombuffer(data,np.uint8,W*H,p).reshape(H,W); p+=W*H
    reg=np.frombuffer(data,np.uint8,(W+16)*(H+16),p).reshape(H+16,W+16); p+=(W+16)*(H+16)
    cmx=np.frombuffer(data,np.uint16,17,p); p+=34
    cmy=np.frombuffer(data,np.uint16,17,p); p+=34
    calls.append((idx,W,mx,my,cost0,ox-mx,oy-my,ocost,f3bc,fe,reg,cmx,cmy))
print(len(calls))
import functools
def run(call, f, base, qtab):
    idx,W,mx,my,cost0,dxo,dyo,ocost,f3bc,fe,reg,cmx,cmy=call
    memo={}
    def M(dx,dy):
        if (dx,dy) not in memo:
            memo[(dx,dy)]=f(had(fe,interp(reg,8,8,W,W,dx,dy),W))+int(cmx[8+dx])+int(cmy[8+dy])
        return memo[(dx,dy)]
    best = M(0,0) if base=='centre' else cost0
    bx=by=0
    for k in range(8):
        c=M(HX[k],HY[k])
        if c<best: best=c; bx,by=HX[k],HY[k]
    cx,cy=bx,by
    for k in range(8):
        c=M(cx+qtab[0][k],cy+qtab[1][k])
        if c<best: best=c; bx,by=cx+qtab[0][k],cy+qtab[1][k]
    return bx,by,best
Q1=([-1,0,1,-1,1,-1,0,1],[-1,-1,-1,0,0,1,1,1])
sub=[c for c in calls if c[8]==1][:500]
for fname,f in (("had",lambda h:h),("had>>1",lambda h:h>>1),("(had+1)>>1",lambda h:(h+1)>>1),("(had+2)>>2",lambda h:(h+2)>>2)):
    for base in ("centre","cost0"):
        ok=sum(1 for c in sub if run(c,f,base,Q1)[:2]==(c[5],c[6]))
        print(fname,base,"mv match",ok,"/",len(sub))
print("---- details for calls with a half-pel result")
shown=0
for c in calls:
    idx,W,mx,my,cost0,dxo,dyo,ocost,f3bc,fe,reg,cmx,cmy=c
    if f3bc!=1 or (dxo,dyo)==(0,0) or dxo%2 or dyo%2 or W!=8: continue
    rows=[]
    for (dx,dy) in [(0,0)]+list(zip(HX,HY)):
        pr=interp(reg,8,8,W,W,dx,dy); h=had(fe,pr,W); s=int(np.abs(fe.astype(int)-pr).sum()); r=int(cmx[8+dx])+int(cmy[8+dy])
        rows.append(((dx,dy),h,s,r))
    print("call",idx,"cost0",cost0,"ref chose",(dxo,dyo),"cost",ocost)
    print("   ", "  ".join(f"{d}:had {h} sad {s} r {r}" for d,h,s,r in rows))
    shown+=1
    if shown>=6: break
print("---- hypothesis: SAD + rate")
def run_sad(call, base, order_q):
    idx,W,mx,my,cost0,dxo,dyo,ocost,f3bc,fe,reg,cmx,cmy=call
    memo={}
    def M(dx,dy):
        if (dx,dy) not in memo:
            memo[(dx,dy)]=int(np.abs(fe.astype(int)-interp(reg,8,8,W,W,dx,dy)).sum())+int(cmx[8+dx])+int(cmy[8+dy])
        return memo[(dx,dy)]
    best = M(0,0) if base=='centre' else cost0
    bx=by=0
    for k in range(8):
        c=M(HX[k],HY[k])
        if c<best: best=c; bx,by=HX[k],HY[k]
    cx,cy=bx,by
    for k in range(8):
        c=M(cx+order_q[0][k],cy+order_q[1][k])
        if c<best: best=c; bx,by=cx+order_q[0][k],cy+order_q[1][k]
    return bx,by,best
allc=[c for c in calls if c[8]==1]
ok=[c for c in allc if run_sad(c,'cost0',Q1)[:2]==(c[5],c[6])]
print("SAD cost0-baseline: mv match",len(ok),"/",len(allc))
bad=[c for c in allc if run_sad(c,'cost0',Q1)[:2]!=(c[5],c[6])]
for c in bad[:8]:
    print("  call",c[0],"W",c[1],"ref",(c[5],c[6]),c[7],"ours",run_sad(c,'cost0',Q1))
okc=sum(1 for c in allc if run_sad(c,'cost0',Q1)[2]==c[7])
print("cost also equal:",okc)
