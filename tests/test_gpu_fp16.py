"""The fp16-operand build (libjat_hip_fp16.so, csrc/jat_dtype.h): `torch.amp.autocast('cuda')` of the v3mod2 trainer is
fp16 with a dynamic loss scale (train_ddp_v3mod2.py:745,854).  The operand dtype is a process-level choice
(JAT_OPERAND_DTYPE=fp16), so this module re-runs the per-kernel, model and training-step parity tests in a child process
against that library: the same reference goldens and the same gates (fp16 rounds operands with 3 more mantissa bits than
bf16), plus the fp16-specific behaviour — operand overflow (> 65504) must surface as a skipped step and a halved scale."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(args, extra_env=None, timeout=900):
    env = dict(os.environ, JAT_OPERAND_DTYPE="fp16", **(extra_env or {}))
    env.pop("JAT_LIB_PATH", None)
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=timeout)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    return tail


def test_fp16_library_is_what_the_env_selects():
    code = "import jatsr_amd._lib as L; print(L.operand_dtype(), L.LIB_PATH)"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, JAT_OPERAND_DTYPE="fp16"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    assert out.stdout.split()[0] == "fp16" and out.stdout.split()[1].endswith("libjat_hip_fp16.so")


def test_fp16_kernels_and_forward_parity():
    _child(["tests/test_gpu_kernels.py"])
    _child(["tests/test_gpu_model.py", "-k", "forward_vs_reference or sampler_vs_reference or benchmarked or fused"])


def test_fp16_training_step_parity_and_loss_scaling():
    _child(["tests/test_gpu_train.py", "-k",
            "train_step_vs_reference_golden or v3mod2 or scale_invariant or non_finite or skipped_step or fp16"])
