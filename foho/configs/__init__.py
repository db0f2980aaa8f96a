"""foho.configs: OptimizationConfig (reference src/foho/configs/guid_config.py:6-32), foho_root() and third_party_root()
(src/foho/configs/paths.py:8-14).  PipelineConfig / load_config belong to the orchestrator (src/foho/main.py), which is
outside the hot path: they are looked up lazily in a FollowMyHold checkout further down sys.path
(foho/configs/pipeline.py there), so `from foho.configs import PipelineConfig, load_config` (main.py:11) keeps working
when this package shadows the checkout's."""
import os
from pkgutil import extend_path

from followmyhold_amd.engine import OptimizationConfig  # noqa: F401

__path__ = extend_path(__path__, __name__)


def foho_root() -> str:
    """Project root (reference paths.py:8-10; this tree has no src/ level: foho/configs -> foho -> root)."""
    return os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def third_party_root() -> str:
    """<project root>/third_party (paths.py:13-14), or $FOHO_THIRD_PARTY when set."""
    return os.environ.get("FOHO_THIRD_PARTY") or os.path.join(foho_root(), "third_party")


def __getattr__(name):
    if name in ("PipelineConfig", "load_config"):
        import importlib
        try:
            mod = importlib.import_module("foho.configs.pipeline")
        except ImportError as e:
            raise AttributeError(f"foho.configs.{name} is part of the FollowMyHold orchestrator (src/foho/configs/"
                                 f"pipeline.py); put the checkout's src/ on sys.path after this repository") from e
        return getattr(mod, name)
    raise AttributeError(f"module 'foho.configs' has no attribute {name!r}")


__all__ = ["OptimizationConfig", "foho_root", "third_party_root", "PipelineConfig", "load_config"]
