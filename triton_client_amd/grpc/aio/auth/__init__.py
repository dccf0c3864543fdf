"""Authentication plugins usable with the asyncio client of this transport."""

from ...._auth import BasicAuth

__all__ = ["BasicAuth"]
