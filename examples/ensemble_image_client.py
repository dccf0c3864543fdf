#!/usr/bin/env python3
"""Sends ENCODED image files (JPEG/PNG bytes) as a BYTES tensor to an ensemble
that decodes, preprocesses and classifies on the server.  (Reference:
``src/python/examples/ensemble_image_client.py``.)"""
import numpy as np
from _common import fail, parser
from _image_example import batches, list_images, print_classes

import tritonclient.grpc as grpcclient
import tritonclient.http as httpclient
from tritonclient.utils import InferenceServerException

if __name__ == "__main__":
    ap = parser(__doc__, 8000, protocol_flag=True)
    ap.add_argument("-m", "--model-name", default="preprocess_inception_ensemble")
    ap.add_argument("-b", "--batch-size", type=int, default=1)
    ap.add_argument("-c", "--classes", type=int, default=1)
    ap.add_argument("image_filename", nargs="?", default=None)
    flags = ap.parse_args()
    if flags.image_filename is None:
        fail("an image file or folder is required")
    mod = grpcclient if flags.protocol.lower() == "grpc" else httpclient
    client = mod.InferenceServerClient(flags.url, verbose=flags.verbose)
    filenames = list_images(flags.image_filename)
    blobs = []
    for name in filenames:
        with open(name, "rb") as f:
            blobs.append(f.read())
    try:
        for sent, batch in enumerate(batches(filenames, flags.batch_size), start=1):
            data = np.array([blobs[i] for i in batch], dtype=np.object_).reshape(len(batch), 1)
            inp = mod.InferInput("INPUT", [len(batch), 1], "BYTES")
            inp.set_data_from_numpy(data)
            out = mod.InferRequestedOutput("OUTPUT", class_count=flags.classes)
            result = client.infer(flags.model_name, [inp], outputs=[out], request_id=str(sent))
            print_classes(result, "OUTPUT", flags.batch_size, True, sent, [filenames[i] for i in batch])
    except InferenceServerException as e:
        fail("inference failed: " + str(e))
    client.close()
    print("PASS")
