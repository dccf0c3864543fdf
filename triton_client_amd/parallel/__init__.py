"""Multi-GPU scale-out: one process (and one model instance) per MI355X.

``fanout``    RCCL collectives (``torch.distributed`` backend "nccl" == RCCL on
              ROCm) that shard a batched request from an ingest GPU across the
              serving GPUs and gather the results back
``direct``    ``DirectPeerExchange``: the same fan-out / gather as plain device-to-
              device copies into peer-mapped (hipIpc) buffers, one stream per
              xGMI link, no collective launches
``group``     ``ShardedModelGroup``: scatter -> local forward -> gather
``endpoints`` client-side request sharding over several server endpoints with
              dead-instance detection (the load generator's multi-GPU mode)
``cluster``   launcher for one v2 server per GPU

The reference has no multi-device code at all (it is a single-process client);
its closest hook is the Java client's pluggable endpoint rotation
(``src/java/.../endpoint/AbstractEndpoint.java:53-67``).
"""

from .direct import DirectPeerExchange  # noqa: F401
from .endpoints import EndpointPool, ShardedClient  # noqa: F401
from .fanout import BatchFanout  # noqa: F401
from .group import ShardedModelGroup  # noqa: F401
