"""foho.configs: OptimizationConfig (reference src/foho/configs/guid_config.py:6-32), foho_root() and third_party_root()
(src/foho/configs/paths.py:8-14).  PipelineConfig / load_config belong to the orchestrator (src/foho/main.py), which is
outside the hot path."""
import os

from followmyhold_amd.engine import OptimizationConfig  # noqa: F401


def foho_root() -> str:
    """Project root (reference paths.py:8-10; this tree has no src/ level: foho/configs -> foho -> root)."""
    return os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def third_party_root() -> str:
    """<project root>/third_party (paths.py:13-14), or $FOHO_THIRD_PARTY when set."""
    return os.environ.get("FOHO_THIRD_PARTY") or os.path.join(foho_root(), "third_party")


__all__ = ["OptimizationConfig", "foho_root", "third_party_root"]
