"""Protobuf request builders shared by the synchronous and asyncio gRPC
clients, so the two differ only in how the RPC is invoked.

Every builder returns ``(rpc_name, request_message)``.  Field usage follows the
reference client (``tritonclient/grpc/_client.py:298-1790``).
"""

import base64
import json

from google.protobuf.json_format import MessageToJson

from ..utils import raise_error
from . import service_pb2 as pb


def check_version(model_version):
    if type(model_version) != str:
        raise_error("model version must be a string")


def to_json(message):
    return json.loads(MessageToJson(message, preserving_proto_field_name=True))


def server_live():
    return "ServerLive", pb.ServerLiveRequest()


def server_ready():
    return "ServerReady", pb.ServerReadyRequest()


def model_ready(model_name, model_version):
    check_version(model_version)
    return "ModelReady", pb.ModelReadyRequest(name=model_name, version=model_version)


def server_metadata():
    return "ServerMetadata", pb.ServerMetadataRequest()


def model_metadata(model_name, model_version):
    check_version(model_version)
    return "ModelMetadata", pb.ModelMetadataRequest(name=model_name, version=model_version)


def model_config(model_name, model_version):
    check_version(model_version)
    return "ModelConfig", pb.ModelConfigRequest(name=model_name, version=model_version)


def repository_index():
    return "RepositoryIndex", pb.RepositoryIndexRequest()


def model_load(model_name, config, files):
    request = pb.RepositoryModelLoadRequest(model_name=model_name)
    if config is not None:
        request.parameters["config"].string_param = config
    if files is not None:
        for path, content in files.items():
            request.parameters[path].bytes_param = content
    return "RepositoryModelLoad", request


def model_unload(model_name, unload_dependents):
    request = pb.RepositoryModelUnloadRequest(model_name=model_name)
    request.parameters["unload_dependents"].bool_param = unload_dependents
    return "RepositoryModelUnload", request


def model_statistics(model_name, model_version):
    check_version(model_version)
    return "ModelStatistics", pb.ModelStatisticsRequest(name=model_name, version=model_version)


def trace_setting(model_name, settings):
    request = pb.TraceSettingRequest()
    if model_name is not None and model_name != "":
        request.model_name = model_name
    for key, value in (settings or {}).items():
        entry = request.settings[key]  # touching the key creates it: empty == clear
        if value is not None:
            entry.value.extend(value if isinstance(value, list) else [value])
    return "TraceSetting", request


def log_settings(settings):
    request = pb.LogSettingsRequest()
    for key, value in (settings or {}).items():
        entry = request.settings[key]
        if value is None:
            continue
        if key in ("log_file", "log_format"):
            entry.string_param = value
        elif key == "log_verbose_level":
            entry.uint32_param = value
        else:
            entry.bool_param = value
    return "LogSettings", request


def sysshm_status(region_name):
    return "SystemSharedMemoryStatus", pb.SystemSharedMemoryStatusRequest(name=region_name)


def sysshm_register(name, key, byte_size, offset):
    return "SystemSharedMemoryRegister", pb.SystemSharedMemoryRegisterRequest(
        name=name, key=key, offset=offset, byte_size=byte_size)


def sysshm_unregister(name):
    return "SystemSharedMemoryUnregister", pb.SystemSharedMemoryUnregisterRequest(name=name)


def devshm_status(region_name):
    return "CudaSharedMemoryStatus", pb.CudaSharedMemoryStatusRequest(name=region_name)


def devshm_register(name, raw_handle, device_id, byte_size):
    # callers hand over the base64 text produced by get_raw_handle(); the wire
    # carries the 64 raw bytes
    return "CudaSharedMemoryRegister", pb.CudaSharedMemoryRegisterRequest(
        name=name, raw_handle=base64.b64decode(raw_handle), device_id=device_id, byte_size=byte_size)


def devshm_unregister(name):
    return "CudaSharedMemoryUnregister", pb.CudaSharedMemoryUnregisterRequest(name=name)
