import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench_slices as bs
from xllm_amd import ops, layers
# monkeypatch forward_experts to capture router logits
seen = []
orig = layers.FusedMoE.forward_experts
def fe(self, x, logits):
    if not seen:
        w, ids = ops.moe_active_topk(logits.reshape(x.size(0), -1), self.topk, self.n_group, self.topk_group, self.renormalize, self.scoring_func == "sigmoid" and 1 or 0, self.correction_bias, self.route_scale) if False else (None, None)
        seen.append(logits.detach().clone())
    return orig(self, x, logits)
layers.FusedMoE.forward_experts = fe
class A: steps=3; warmup=1; no_pmc=True
r = bs.cfg4_slice(A(), "cuda")
lg = seen[0].float()
print("router logits", lg.shape, "std over tokens of each expert's logit (mean):", lg.std(0).mean().item(), " std over experts (mean):", lg.std(1).mean().item())
top = lg.topk(8, dim=1).indices
cnt = torch.bincount(top.flatten(), minlength=256)
print("plain top-8 (no groups/bias): active experts", int((cnt > 0).sum()), "max rows", int(cnt.max()))
