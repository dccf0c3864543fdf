"""Authentication plugins usable with this transport."""

from ..._auth import BasicAuth

__all__ = ["BasicAuth"]
