import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import jatsr_amd, jatsr_amd.recipe as recipe, jatsr_amd._lib as L
print("dtype", L.operand_dtype())
cfg = recipe.CONFIGS["v3mod2"]
m = jatsr_amd.JaT_AudioSR_V3(**cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg).items()}, strict=False)
m = m.cuda().eval()
temb = torch.from_numpy(recipe.gaussian("blk_t", (2, 1280), 2)).cuda()
x = torch.from_numpy(recipe.gaussian("blk_x", (2, 128, 1280), 1)).cuda()
def run(f): os.environ["JAT_FUSE_QKV_ATTN"] = f; return m.blocks[0](x, temb)
a1, a2, b1, b2 = run("2"), run("2"), run("0"), run("0")
print("fused==fused", torch.equal(a1, a2), " sep==sep", torch.equal(b1, b2), " fused==sep", torch.equal(a1, b1), float((a1 - b1).abs().max()))
d = (a1 - b1).abs()
print("rows differing per sample:", [(int((d[b].amax(-1) > 0).sum())) for b in range(2)], " cols differing:", int((d.amax((0, 1)) > 0).sum()))
