"""Multi-GPU scale-out: one process (and one model instance) per MI355X.

``sharded``   ``ShardRoot`` / ``ShardWorker``: one batch split over replica processes (one per
              GPU) as a servable execution path (model ``resnet50_sharded``).  Shards leave the
              ingest GPU as peer stores into hipIpc-mapped inboxes (all xGMI links at once) or
              as RCCL point-to-point operations; hand-offs are flags the GPUs set and poll -
              no control message and no process-group call per step
``fanout``    RCCL collectives (``torch.distributed`` backend "nccl" == RCCL on ROCm) that
              shard a batched tensor from an ingest rank across the ranks and gather results
``endpoints`` client-side request sharding over several server endpoints with
              dead-instance detection (the load generator's multi-GPU mode)
``cluster``   launcher for one v2 server per GPU

The reference has no multi-device code at all (it is a single-process client);
its closest hook is the Java client's pluggable endpoint rotation
(``src/java/.../endpoint/AbstractEndpoint.java:53-67``).
"""

from .endpoints import EndpointPool, ShardedClient  # noqa: F401
from .fanout import BatchFanout  # noqa: F401
from .sharded import ShardRoot, ShardWorker  # noqa: F401
