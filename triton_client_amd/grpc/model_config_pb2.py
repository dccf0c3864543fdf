"""``inference.ModelConfig`` and friends (built at import time, see ``_pb``)."""

from . import _pb

DESCRIPTOR = _pb.FILES["model_config.proto"]

globals().update(_pb.message_classes("model_config.proto"))
for _name, _wrapper in _pb.enum_wrappers("model_config.proto").items():
    globals()[_name] = _wrapper
    # protoc also hoists the values of top-level enums into the module
    for _value in _wrapper.DESCRIPTOR.values:
        globals()[_value.name] = _value.number
del _name, _wrapper, _value
