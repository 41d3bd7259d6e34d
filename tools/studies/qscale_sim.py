"""Numerical study (CPU, NumPy): what would folding the softmax scale into Q cost in accuracy?

The tuned kernel spends one fused multiply-add per score on  s * c - m  (c = log2(e)/sqrt(D)) before the
exponential.  Pre-multiplying the fp16 Q rows by c (and feeding -m through the first MFMA's C operand)
would remove it, but rounds Q * c to fp16.  This script models both pipelines (fp32 scores, fp16 P, fp32 row
sum, fp16 output) against an fp64 softmax on N(0, amp^2) data and prints max-abs and relative-L2 errors.
Result (kept in the design notes): the error grows about 3.5x on N(0,1) data and passes the 2e-3 max-abs gate
only there; at amp = 2 and 3 (peaked softmax) it is 2.7e-3 and 6.7e-3.  The fold was therefore not built.
"""
import numpy as np
rng=np.random.default_rng(0)
def run(L,D=128,amp=1.0,H=2):
    c=np.float32(1.4426950408889634/np.sqrt(D))
    res=[]
    for h in range(H):
        q=(rng.standard_normal((L,D))*amp).astype(np.float16); k=(rng.standard_normal((L,D))*amp).astype(np.float16); v=rng.standard_normal((L,D)).astype(np.float16)
        s64=(q.astype(np.float64)@k.astype(np.float64).T)/np.sqrt(D)
        p=np.exp(s64-s64.max(1,keepdims=True)); ref=(p/p.sum(1,keepdims=True))@v.astype(np.float64)
        out=[]
        for mode in (0,1):
            if mode==0:
                s=(q.astype(np.float32)@k.astype(np.float32).T)*c
            else:
                qs=(q.astype(np.float32)*c).astype(np.float16)
                s=(qs.astype(np.float32)@k.astype(np.float32).T)
            m=s.max(1,keepdims=True)
            pf=np.exp2(s-m).astype(np.float32)
            l=pf.sum(1,keepdims=True)
            ph=pf.astype(np.float16).astype(np.float32)
            o=((ph@v.astype(np.float32))/l).astype(np.float16).astype(np.float64)
            out.append((np.abs(o-ref).max(), np.linalg.norm(o-ref)/np.linalg.norm(ref)))
        res.append(out)
    r=np.array(res).max(0)
    print(f"L={L} amp={amp}: base maxabs {r[0,0]:.2e} relL2 {r[0,1]:.2e} | prescaled maxabs {r[1,0]:.2e} relL2 {r[1,1]:.2e}")
for L in (512,2048): run(L)
run(512,amp=2.0); run(512,amp=3.0)
