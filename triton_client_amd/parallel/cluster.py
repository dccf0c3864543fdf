"""Launch one v2 server process per GPU (the serving side of the scaling runs).

Each instance is an independent ``python -m triton_client_amd.server --gpu i``
child bound to its own ports; clients shard requests over the instances with
:class:`..endpoints.ShardedClient` or the load generator's ``--urls`` option.
"""

from __future__ import annotations

from ..server import spawn_server


class GpuCluster:
    def __init__(self, gpus, models=("resnet50",), env=None, extra_args=()):
        self.gpus = list(gpus)
        self.models = list(models)
        self.env = env
        self.extra_args = tuple(extra_args)
        self.servers = []

    def start(self):
        try:
            for g in self.gpus:
                self.servers.append(spawn_server(models=self.models, gpu=g, env=self.env, extra_args=self.extra_args))
        except Exception:
            self.stop()
            raise
        return self

    @property
    def grpc_urls(self):
        return [s.grpc_url for s in self.servers]

    @property
    def http_urls(self):
        return [s.http_url for s in self.servers]

    def stop(self):
        for s in self.servers:
            s.stop()
        self.servers = []

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.stop()
