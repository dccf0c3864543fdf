#!/usr/bin/env python3
"""Model repository control over GRPC: index, load / unload, config
override, unknown-model failure.  (Reference:
``src/python/examples/simple_grpc_model_control.py``.)"""
import json

from _common import fail, parser

import tritonclient.grpc as client_module
from tritonclient.utils import InferenceServerException

if __name__ == "__main__":
    flags = parser(__doc__, 8001).parse_args()
    client = client_module.InferenceServerClient(url=flags.url, verbose=flags.verbose)
    index = client.get_model_repository_index()
    count = len(index) if isinstance(index, list) else len(index.models)
    print(index)
    if count < 8:
        fail("expected at least 8 models in the repository index")
    client.unload_model("simple")
    if client.is_model_ready("simple"):
        fail("FAILED : Unload Model")
    client.load_model("simple")
    if not client.is_model_ready("simple"):
        fail("FAILED : Load Model")
    client.load_model("simple", config=json.dumps({"max_batch_size": 16, "backend": "python"}))
    config = client.get_model_config("simple")
    max_batch = config["max_batch_size"] if isinstance(config, dict) else config.config.max_batch_size
    if max_batch != 16:
        fail("FAILED : config override")
    client.load_model("simple")
    try:
        client.load_model("wrong_model_name")
        fail("expected a load failure")
    except InferenceServerException as ex:
        if "failed to load" not in ex.message():
            fail("FAILED : Load Model wrong_model_name: " + str(ex))
    client.close()
    print("PASS: model control")
