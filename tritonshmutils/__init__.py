"""Deprecated alias package: ``tritonshmutils.shared_memory`` and
``tritonshmutils.cuda_shared_memory`` forward to ``tritonclient.utils.*``."""

import warnings

warnings.simplefilter("always", DeprecationWarning)
warnings.warn(
    "The package `tritonshmutils` is deprecated and will be removed in a future version. Please use instead "
    "`tritonclient.utils`",
    DeprecationWarning,
)
