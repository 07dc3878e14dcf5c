"""Randomised end-to-end parity sweep (GPU box): random picture sizes (8..472 x 8..312), QP 0..51, DIA/HEX/UMH, search range, sub-pel / deblock /
SAO switches, and the three GOP structures (IPPP, multi-reference P, hierarchical B): every reconstructed picture of the HIP pipeline must equal the
oracle's.  Not collected by pytest (no test_ prefix): run `python tests/fuzz_parity.py SEED COUNT` through gpurun.  Round 1: seeds 1..4, 300 cases, 0 failures.  Round 6: half of the pyramid cases run the host's tool set with the tools of every B picture drawn at random
(ks265_frame_set_picture_tools)."""
import sys, itertools, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from oracle_lib import OraclePipeline
from ks265codec_amd.lib import KsContext, KsFrame
from ks265codec_amd.synth import make_clip, lambda_q4
from ks265codec_amd.gop import hier_order
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
ks=KsContext(0)
nfail=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 30):
    W=int(rng.integers(1,60))*8; H=int(rng.integers(1,40))*8
    qp=int(rng.integers(0,52)); me=int(rng.integers(0,3)); rangev=int(rng.choice([8,16,32,64])); subme=int(rng.integers(0,2)); df=int(rng.integers(0,2)); sao=int(rng.integers(0,2))
    mode=rng.choice(["ippp","mref","hier"])
    n=6
    clip=make_clip(W,H,9,seed=int(rng.integers(0,10000)),noisy=bool(rng.integers(0,2)))
    kw=dict(me_range=rangev,subme=subme,deblock=df,sao=sao,me_method=me)
    lean_mix=mode=="hier" and bool(rng.integers(0,2))     # round 6: the host's tool set with tools switched per B picture (ks265_frame_set_picture_tools: lean / near-lean / full / + interMeHex)
    if lean_mix: kw.update(sdh=1,pre_search=1,merge=1,bi_refine=int(rng.choice([0,2])),rdo=4,intra_inter=1,propagate=1,skip_rd=1,sao=1,me_hex_thr=16 if me==2 else 0)
    o=OraclePipeline(W,H,qp,lambda_q4(qp),**kw)
    try:
        with KsFrame(ks,W,H,qp,lambda_q4(qp),bframes=3,refs=3,**kw) as f:
            src=f.new_pic()
            if mode=="hier":
                G=4; dg=[f.new_pic() for _ in range(G+1)]; do={}
                for d,kind,r0,r1,layer in itertools.islice(hier_order(G,128),2*G+1):
                    q=min(51,qp if kind=="I" else qp+1+layer)
                    o.set_qp(q,lambda_q4(q)); f.set_qp(q,lambda_q4(q))
                    if lean_mix:
                        t=[(-1,-1,-1,-1),(0,0,0,-1),(0,-1,0,-1),(0,0,0,1 if me==2 else -1)][int(rng.integers(0,4))] if kind=="B" else (-1,-1,-1,-1)
                        o.set_picture_tools(*t); f.set_picture_tools(*t)
                    do[d]=o.encode(clip[d],kind,do.get(r0),do.get(r1))
                    f.load_i420(ks.dev(clip[d]),src); out=dg[d%(G+1)]
                    if kind=="B": f.encode_picture_b(src,dg[r0%(G+1)],dg[r1%(G+1)],out)
                    else: f.encode_picture(src,dg[r0%(G+1)] if r0 is not None else out,kind=="I",out)
                    got,exp=ks.host(f.store_i420(out),np.uint8),o.store(do[d])
                    assert (got==exp).all(),(d,kind,int((got!=exp).sum()))
            else:
                dpo,dpg=[],[]
                for t in range(n):
                    f.load_i420(ks.dev(clip[t]),src); out=f.new_pic()
                    if t==0:
                        eo=o.encode(clip[0],"I"); f.encode_picture(src,out,True,out)
                    elif mode=="mref":
                        eo=o.encode_mref(clip[t],dpo[:3]); f.encode_picture_mref(src,dpg[:3],out)
                    else:
                        eo=o.encode(clip[t],"P",dpo[0]); f.encode_picture(src,dpg[0],False,out)
                    got,exp=ks.host(f.store_i420(out),np.uint8),o.store(eo)
                    assert (got==exp).all(),(t,int((got!=exp).sum()))
                    dpo.insert(0,eo); dpg.insert(0,out)
        print("ok",W,H,qp,me,rangev,subme,df,sao,mode,"lean-mix" if lean_mix else "")
    except AssertionError as e:
        nfail+=1; print("FAIL",W,H,qp,me,rangev,subme,df,sao,mode,e)
print("failures",nfail)
