"""CPU tests for the data formats either side of the path (jatsr_amd.io) and the inference driver's arguments:
latent container, the three normalisation-statistics formats of infer_test_v3m2.py:300-332, chunk constants."""
import json

import numpy as np
import pytest
import torch

import jatsr_amd.io as jio
from jatsr_amd.infer import build_parser


def test_latent_file_roundtrip(tmp_path):
    hr, lr = torch.randn(32, 50), torch.randn(32, 50)
    p = tmp_path / "clip.pt"
    jio.save_latent_file(p, hr_latent=hr, lr_latent=lr, metadata={"sr": 44100})
    raw = torch.load(p, weights_only=False)
    assert raw["hr_latent"].dtype == torch.float16 and raw["lr_latent"].shape == (32, 50)   # prepare_dataset_v5.py:255-264
    hr2, lr2 = jio.load_latent_file(p)
    assert hr2.dtype == torch.float32 and torch.allclose(lr2, lr.half().float())
    torch.save({"hr_latent": hr.half()}, tmp_path / "bad.pt")
    with pytest.raises(KeyError):
        jio.load_latent_file(tmp_path / "bad.pt")


def test_stats_formats(tmp_path):
    C = 8
    st = {k: torch.rand(C) + 0.5 for k in ("hr_mean", "hr_std", "lr_mean", "lr_std")}
    (tmp_path / "s.json").write_text(json.dumps({k: v.tolist() for k, v in st.items()}))
    a = jio.load_stats(str(tmp_path / "s.json"), channels=C)
    torch.save(st, tmp_path / "s.pt")
    b = jio.load_stats(str(tmp_path / "s.pt"), channels=C)
    for k in st:
        assert torch.allclose(a[k], st[k]) and torch.allclose(b[k], st[k])
    # running sums over [HR ; LR] channels (infer_test_v3m2.py:316-326)
    x = torch.randn(1000, 2 * C).double() * 2 + 1
    torch.save({"sum": x.sum(0), "sq_sum": (x * x).sum(0), "count": 1000}, tmp_path / "run.pt")
    c = jio.load_stats(str(tmp_path / "run.pt"), channels=C)
    assert torch.allclose(c["hr_mean"], x[:, :C].mean(0).float(), atol=1e-5)
    assert torch.allclose(c["lr_std"], torch.sqrt(x[:, C:].var(0, unbiased=False) + 1e-8).float(), atol=1e-5)
    torch.save({"foo": 1}, tmp_path / "unk.pt")
    with pytest.raises(ValueError):
        jio.load_stats(str(tmp_path / "unk.pt"), channels=C)
    with pytest.raises(ValueError):
        jio.load_stats(str(tmp_path / "s.json"), channels=C + 1)


def test_chunk_constants_and_cli_flags():
    assert jio.frames_for_seconds(16.0) == 1378 and jio.frames_for_seconds(2.0) == 172   # infer_test_v3m2.py:345-346
    a = build_parser().parse_args(["--checkpoint", "c.pt", "--input-file", "x.pt", "--cfg-scale", "3.0", "--steps", "25",
                                   "--total-seconds", "10", "--stats-file", "s.json", "--val-dir", "v", "--output-dir", "o"])
    assert (a.steps, a.cfg_scale, a.total_seconds, a.device) == (25, 3.0, 10.0, "cuda")   # flags of :237-256
    d = build_parser().parse_args([])
    assert d.steps == 50 and d.cfg_scale == 1.0 and d.checkpoint == "checkpoints/v3_full_run/last.pt"


def test_first_latent_file(tmp_path):
    with pytest.raises(FileNotFoundError):
        jio.first_latent_file(tmp_path)
    for n in ("b.pt", "a.pt", "c.txt"):
        (tmp_path / n).write_bytes(b"")
    assert jio.first_latent_file(tmp_path).endswith("a.pt")
