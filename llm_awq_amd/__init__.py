"""llm_awq_amd -- MI355X-native (gfx950) W4A16 fused dequant+matmul behind llm-awq's WQLinear.

Only the hot path lives here: the `awq_inference_engine` extension API, the
`WQLinear` module mirror, the checkpoint repacker and the K-sharded tensor-parallel wrapper.
"""
from __future__ import annotations

import importlib
import os
import sys

__all__ = ["load_engine", "install_as_awq_inference_engine"]

_EXT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ext")
_engine = None


def load_engine():
    """Import the compiled `awq_inference_engine` torch extension (HIP kernels). No fallback:
    raises ImportError with build instructions if it has not been built."""
    global _engine
    if _engine is None:
        if _EXT_DIR not in sys.path:
            sys.path.insert(0, _EXT_DIR)
        import torch  # noqa: F401  (libtorch must be loaded before the extension)
        try:
            _engine = importlib.import_module("awq_inference_engine")
        except ImportError as e:  # pragma: no cover - exercised on unbuilt trees
            raise ImportError(
                "awq_inference_engine (MI355X HIP extension) is not built: run `python -m llm_awq_amd.build`. "
                "There is deliberately no CPU/PyTorch fallback for the WQLinear hot path.") from e
        if not hasattr(_engine, "gemv_forward_cuda_new") or not hasattr(_engine, "abi_version"):
            raise ImportError("a foreign `awq_inference_engine` module shadows the MI355X build")
    return _engine


def install_as_awq_inference_engine():
    """Make `import awq_inference_engine` (as the reference's qmodule.py:4, fused_mlp.py:8 do) resolve
    to the MI355X build."""
    eng = load_engine()
    sys.modules["awq_inference_engine"] = eng
    return eng
