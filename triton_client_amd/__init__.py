"""MI355X-native inference client stack (KServe-v2 / Triton protocol).

Sub-packages
------------
utils      tensor codec, DLPack shim, system shm, ``hip_shared_memory`` (hipshm)
http/grpc  ``InferenceServerClient`` (sync + asyncio) for both transports
ops        hand-written CDNA4 (gfx950) HIP kernels: fused image preprocessing,
           BYTES pack/unpack, bf16, classification head
models     model zoo served by the in-repo v2 server (ResNet-50, add_sub, ...)
server     minimal v2-protocol server, one PyTorch-ROCm instance per GPU
perf       perf_analyzer-style load generator
parallel   RCCL (torch.distributed "nccl") fan-out / gather across GPUs

The same API is importable under the historical ``tritonclient`` name.
"""

__version__ = "0.1.0"
