import torch, time, sys
sys.path.insert(0,'/root/repo')
import ac_tsr_amd as A
from ac_tsr_amd import _lib
lib=_lib.load()
DEV='cuda'
def run(B,L,H,nh,which,iters=30):
    lib.acattn_select_backward_kernel(which)
    g=torch.Generator().manual_seed(1)
    mk=lambda *s: torch.randn(*s,generator=g).to(DEV)
    t={k:mk(B,L,H).requires_grad_(True) for k in ("q","k","v","qa","ka")}
    t["gl"]=mk(B,L,L).requires_grad_(True)
    dh=H//nh
    w={k:(0.3*torch.randn(*s,generator=g)).to(DEV).requires_grad_(True) for k,s in (("w_order",(1,2*dh)),("b_order",(1,)),("w_dist",(1,2*dh)),("b_dist",(1,)),("scalar",(1,)))}
    lens=torch.randint(1,L+1,(B,),generator=g)
    kv=(torch.arange(L)[None,:]<lens[:,None]).to(torch.uint8).to(DEV)
    mask=A.StructuredMask(kv,causal=True)
    cfg=A.AttentionConfig(n_heads=nh,combine_option="gate")
    out=A.calibrated_attention(t["q"],t["k"],t["v"],t["qa"],t["ka"],t["gl"],mask,cfg,p_drop=0.5,seed=7,**w)
    cot=[mk(B,L,H),mk(B,L,H),mk(B,nh,L,L)]
    loss=sum((o*c).sum() for o,c in zip(out[:3],cot))
    ins=list(t.values())+list(w.values())
    for _ in range(3): torch.autograd.grad(loss,ins,retain_graph=True)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(iters): torch.autograd.grad(loss,ins,retain_graph=True)
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/iters*1e6
import os
shapes = eval(os.environ.get("SHAPES", "((512,50,64,2),(512,64,128,2),(512,200,64,2))"))
for shape in shapes:
    for which,name in ((2,'row'),(1,'stream')):
        if shape[1]>64 and which==2: iters=3
        else: iters=30
        print(shape,name,"%.0f us (autograd node incl. reductions)"%run(*shape,which,iters))
