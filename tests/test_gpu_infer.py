"""End-to-end inference driver on the GPU (the counterpart of reference infer_test_v3m2.py:main): reference-format
checkpoint (+ torch.compile / DDP prefixes) -> latent file -> stats JSON -> chunk plan -> batched CFG sampling ->
crossfade -> output file; checked against the CPU oracle run chunk by chunk."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import jatsr_amd.io as jio  # noqa: E402
import jatsr_amd.recipe as recipe  # noqa: E402
from helpers import rel_l2  # noqa: E402
from jatsr_amd.infer import main as infer_main  # noqa: E402
from oracle import jat_oracle as O  # noqa: E402


def test_infer_cli_end_to_end(tmp_path):
    cfg = recipe.CONFIGS["micro"]
    C, T = cfg["input_channels"], 1500                       # 2 chunks: [0:1378], [1206:1500]
    sd = recipe.make_state_dict(cfg)
    ckpt = {"model_state_dict": {"_orig_mod." + k: torch.from_numpy(v) for k, v in sd.items()},
            "config": dict(cfg, dropout=0.1, drop_path_rate=0.05), "epoch": 1, "global_step": 10}
    torch.save(ckpt, tmp_path / "last.pt")
    lr = recipe.gaussian("cli_lr", (C, T), 1) * 1.5 + 0.2
    hr = recipe.gaussian("cli_hr", (C, T), 2)
    jio.save_latent_file(tmp_path / "clip.pt", hr_latent=torch.from_numpy(hr), lr_latent=torch.from_numpy(lr))
    stats = {"hr_mean": (recipe.gaussian("m1", (C,), 1) * 0.1).tolist(), "hr_std": (np.abs(recipe.gaussian("s1", (C,), 2)) + 0.5).tolist(),
             "lr_mean": (recipe.gaussian("m2", (C,), 3) * 0.1).tolist(), "lr_std": (np.abs(recipe.gaussian("s2", (C,), 4)) + 0.5).tolist()}
    (tmp_path / "stats.json").write_text(json.dumps(stats))
    out = infer_main(["--checkpoint", str(tmp_path / "last.pt"), "--input-file", str(tmp_path / "clip.pt"),
                      "--stats-file", str(tmp_path / "stats.json"), "--output-dir", str(tmp_path / "out"),
                      "--steps", "4", "--cfg-scale", "2.0", "--seed", "7"])
    res = torch.load(out, weights_only=False)
    gen = res["generated_latent"].float().numpy()
    assert gen.shape == (C, T) and np.isfinite(gen).all()
    assert res["metadata"]["frames"] == T and res["lr_latent"].shape == (C, T)

    # oracle: same noise (seeded the same way), chunk by chunk, fp32
    g = torch.Generator(device="cpu").manual_seed(7)
    plan = O.chunk_plan(T)
    assert plan == [(0, 1378), (1206, 1500)]
    noise = [torch.randn(1, C, b - a, generator=g).numpy() for a, b in plan]
    orc = O.OracleModel(cfg, sd, "rms", np.float32)
    lr16 = torch.from_numpy(lr).half().float().numpy()      # the file stores fp16 (prepare_dataset_v5.py:255-264)
    lm, ls = np.asarray(stats["lr_mean"], np.float32)[None, :, None], np.asarray(stats["lr_std"], np.float32)[None, :, None]
    hm, hs = np.asarray(stats["hr_mean"], np.float32)[None, :, None], np.asarray(stats["hr_std"], np.float32)[None, :, None]
    chunks = []
    for (a, b), z0 in zip(plan, noise):
        c = (lr16[None, :, a:b] - lm) / ls
        chunks.append(O.flow_matching_sample(orc, c, z0, 4, 2.0) * hs + hm)
    ref = O.crossfade_chunks(chunks, 172)[0]
    assert rel_l2(gen, ref) < 3e-2
