import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import jatsr_amd, jatsr_amd.recipe as recipe, jatsr_amd._lib as L
from helpers import load_golden, fwd_inputs, rel_l2, sub
print("dtype", L.operand_dtype())
z, meta = load_golden("fwd_v3mod2_T512")
cfg, x_t, t, x_c = fwd_inputs(meta)
m = jatsr_amd.JaT_AudioSR_V3(**cfg)
m.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.make_state_dict(cfg, "rms", meta["salt"]).items()}, strict=False)
m = m.cuda().eval()
c = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
os.environ["JAT_FUSE_QKV_ATTN"] = "2"
f = m(c(x_t), c(t), c(x_c))
os.environ["JAT_FUSE_QKV_ATTN"] = "0"
s = m(c(x_t), c(t), c(x_c))
d = (f - s).abs()
print("equal", torch.equal(f, s), "max diff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel(), "finite", bool(torch.isfinite(f).all()), bool(torch.isfinite(s).all()))
print("fused rel", rel_l2(sub(f.cpu().numpy(), *meta["s_out"]), z["out64"]), "sep rel", rel_l2(sub(s.cpu().numpy(), *meta["s_out"]), z["out64"]))
