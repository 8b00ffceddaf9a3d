import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from bench import synthetic_batch
from hpcs_amd.loss.ultrametric_loss import MetricHyperbolicLoss
dev='cuda'
pts,cat,tgt = synthetic_batch(32,1024,50,16,seed=0)
lab = tgt.reshape(-1); M,E=lab.numel(),32
torch.manual_seed(0)
cent = torch.randn(50, E)
x=(cent[lab] + 1.6 * torch.randn(M,E)).to(dev)
crit=MetricHyperbolicLoss(t_per_anchor=50,fraction=1.2,scale=torch.nn.Parameter(torch.tensor([1e-3])),num_class=50,embedding_size=E,cosface=True,miner=True,rng="device",fused_filter=True).to(dev)
plan=crit.make_plan(lab); labd=lab.to(dev); xg=x.clone().requires_grad_(True)
if os.environ.get("PMC_NOGRAD"):
    with torch.no_grad():
        for _ in range(4): crit.compute_hyp(x,labd,plan=plan)
else:
    for _ in range(4): crit.compute_hyp(xg,labd,plan=plan)
torch.cuda.synchronize()
print("kept", int(crit.last_kept))
