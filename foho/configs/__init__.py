"""foho.configs: OptimizationConfig (reference src/foho/configs/guid_config.py:6-32) and third_party_root()
(src/foho/configs/paths.py:8-14)."""
import os

from followmyhold_amd.engine import OptimizationConfig  # noqa: F401


def third_party_root() -> str:
    """<project root>/third_party, or $FOHO_THIRD_PARTY when set."""
    env = os.environ.get("FOHO_THIRD_PARTY")
    if env:
        return env
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(os.path.dirname(os.path.dirname(here)), "third_party")
