#!/usr/bin/env python3
"""Health, metadata and configuration queries over HTTP.
(Reference: ``src/python/examples/simple_http_health_metadata.py``.)"""
from _common import fail, headers_of, parser

import tritonclient.http as client_module
from tritonclient.utils import InferenceServerException

if __name__ == "__main__":
    flags = parser(__doc__, 8000).parse_args()
    client = client_module.InferenceServerClient(url=flags.url, verbose=flags.verbose)
    headers = headers_of(flags) or {"test": "1"}
    if not client.is_server_live(headers=headers):
        fail("FAILED : is_server_live")
    if not client.is_server_ready():
        fail("FAILED : is_server_ready")
    if not client.is_model_ready("simple"):
        fail("FAILED : is_model_ready")
    metadata = client.get_server_metadata()
    name = metadata["name"] if isinstance(metadata, dict) else metadata.name
    if name != "triton":
        fail("FAILED : get_server_metadata")
    print(metadata)
    model_metadata = client.get_model_metadata("simple", headers=headers)
    model_name = model_metadata["name"] if isinstance(model_metadata, dict) else model_metadata.name
    if model_name != "simple":
        fail("FAILED : get_model_metadata")
    print(model_metadata)
    config = client.get_model_config("simple")
    config_name = config["name"] if isinstance(config, dict) else config.config.name
    if config_name != "simple":
        fail("FAILED: get_model_config")
    try:
        client.get_model_metadata("wrong_model_name")
        fail("expected an error for an unknown model")
    except InferenceServerException as ex:
        if "Request for unknown model" not in ex.message():
            fail("FAILED : get_model_metadata wrong_model_name: " + str(ex))
    client.close()
    print("PASS: health metadata")
