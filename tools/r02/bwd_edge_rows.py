import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from samplenerfro_amd import _lib, ops, synthetic as syn
dev="cuda:0"
pf = torch.from_numpy(syn.init_params_flat(3, fine=False, bias_scale=0.1)["coarse_mlp"]).to(dev)
packed = ops.nerfmlp_pack(pf, _lib.PREC_F16X3)
ok=True
for B,S in ((1,1),(31,1),(33,1),(5,51),(257,1),(64,5)):
    g = torch.Generator(device=dev).manual_seed(B*7+S)
    pd = torch.rand((S,B,4),device=dev,generator=g)*2-1
    dr = torch.nn.functional.normalize(torch.randn((S,B,4),device=dev,generator=g),dim=-1)
    d_raw = torch.randn((S,B,4),device=dev,generator=g)
    res={}
    for mode in ("f32","tf32","bf16"):
        BW=_lib.BACKWARDS[mode]
        pbwd=ops.nerfmlp_pack_bwd(pf,None,BW)
        raw,save=ops.nerfmlp_forward_train(packed,_lib.PREC_F16X3,pd,dr,None,S,B,BW)
        res[mode]=ops.nerfmlp_backward(pbwd,packed,_lib.PREC_F16X3,save,d_raw,S*B,backward=BW).double()
    sc=res["f32"].abs().max()
    e1=float((res["tf32"]-res["f32"]).abs().max()/sc); e2=float((res["bf16"]-res["f32"]).abs().max()/sc)
    fin=all(torch.isfinite(v).all() for v in res.values())
    print(f"rows {B*S:5d}: tf32 vs f32 {e1:.2e}  bf16 vs f32 {e2:.2e} finite {fin}")
    ok = ok and fin and e1<5e-3 and e2<5e-2
print("OK" if ok else "FAIL")
