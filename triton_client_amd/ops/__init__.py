"""Hand-written CDNA4 (gfx950) HIP kernels and their Python entry points.

================  ==========================================================
``preprocess``    fused image resize + colour convert + normalise + layout +
                  cast, writing straight into a hipshm region
``codec``         BYTES tensor pack / index / gather, BF16 <-> FP32
``classification`` top-k and ``"value:index:label"`` payload formatting
``head``          MFMA fully connected classification head (fp16)
================  ==========================================================

All kernels live in ``csrc/hip`` and are compiled into ``libtca_hip.so`` for
gfx950 only; there is no fallback path: importing an op on a machine whose
library cannot be built raises.
"""
