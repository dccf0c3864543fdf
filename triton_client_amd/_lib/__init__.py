"""Holds the shared objects produced by ``triton_client_amd._build`` (git-ignored)."""
