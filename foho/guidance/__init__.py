"""Hot-path modules of foho.guidance; anything else resolves to a FollowMyHold checkout further down sys.path."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
