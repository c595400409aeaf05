import os, sys, torch
sys.path.insert(0, os.getcwd())
from transformer_mm_explainability_amd import ops
torch.manual_seed(0)
for (L,B,H,N) in [(2,1,2,197),(2,1,2,577)]:
    attn=[torch.rand(B*H,N,N,device="cuda").softmax(-1) for _ in range(L)]
    grad=[torch.randn(B*H,N,N,device="cuda")*0.05 for _ in range(L)]
    ops.set_option("self_chain_rows",0); want=ops.relevancy_self_chain(attn,grad,B).clone()
    ops.set_option("self_chain_rows",1); got=ops.relevancy_self_chain(attn,grad,B).clone()
    err=(got-want).abs()[0]
    colerr=err.max(0).values
    bad=(colerr>1e-6).nonzero().flatten().tolist()
    print(N,"bad columns:",len(bad),bad[:24],"...",bad[-8:])
    rowerr=err.max(1).values; print("  bad rows:", int((rowerr>1e-6).sum()))
    # is got[:, c] == want[:, c'] for some c'?
    for c in bad[:4]:
        d=(want[0]-got[0][:,c:c+1]).abs().max(0).values
        print("  got col",c,"closest want col",int(d.argmin()),float(d.min()))
