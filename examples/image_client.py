#!/usr/bin/env python3
"""Image classification client: reads the model's metadata to learn its input
geometry, preprocesses images (on the host, or on the GPU straight into device
shared memory), sends batches synchronously / asynchronously / over a gRPC
stream and prints the top-K classes.  (Reference:
``src/python/examples/image_client.py``.)"""
import queue
from functools import partial

from _common import fail, parser
from _image_example import HipInputStage, batches, decode, host_preprocess, list_images, parse_model, print_classes

import numpy as np
import tritonclient.grpc as grpcclient
import tritonclient.http as httpclient
from tritonclient.utils import InferenceServerException


def completion(done, result, error):
    done.put((result, error))


if __name__ == "__main__":
    ap = parser(__doc__, 8000, protocol_flag=True)
    ap.add_argument("-a", "--async", dest="async_set", action="store_true", help="Use asynchronous inference API")
    ap.add_argument("--streaming", action="store_true", help="Use streaming inference API (gRPC only)")
    ap.add_argument("-m", "--model-name", required=True)
    ap.add_argument("-x", "--model-version", default="")
    ap.add_argument("-b", "--batch-size", type=int, default=1)
    ap.add_argument("-c", "--classes", type=int, default=1, help="Number of class results to report")
    ap.add_argument("-s", "--scaling", choices=["NONE", "INCEPTION", "VGG"], default="NONE")
    ap.add_argument("--preprocess", choices=["host", "hip"], default="host",
                    help="host: numpy/Pillow on the CPU; hip: fused kernel into device shared memory")
    ap.add_argument("-g", "--device", type=int, default=0, help="GPU ordinal for --preprocess hip")
    ap.add_argument("--ring-depth", type=int, default=4,
                    help="--preprocess hip: batch slots in the input region (requests in flight while the next is "
                         "filled)")
    ap.add_argument("image_filename", nargs="?", default=None, help="Input image / input folder")
    flags = ap.parse_args()
    grpc_mode = flags.protocol.lower() == "grpc"
    if flags.streaming and not grpc_mode:
        fail("Streaming is only allowed with gRPC protocol")
    if flags.image_filename is None:
        fail("an image file or folder is required")
    mod = grpcclient if grpc_mode else httpclient
    try:
        client = (grpcclient.InferenceServerClient(flags.url, verbose=flags.verbose) if grpc_mode else
                  httpclient.InferenceServerClient(flags.url, verbose=flags.verbose,
                                                   concurrency=20 if flags.async_set else 1))
        metadata = client.get_model_metadata(flags.model_name, flags.model_version)
        config = client.get_model_config(flags.model_name, flags.model_version)
    except InferenceServerException as e:
        fail("failed to query the model: " + str(e))
    spec = parse_model(metadata, config)
    batched = spec.max_batch_size > 0
    if not batched and flags.batch_size != 1:
        fail("ERROR: This model doesn't support batching.")

    filenames = list_images(flags.image_filename)
    pixels = [decode(f, spec.c) for f in filenames]
    stage = HipInputStage(client, spec, flags.scaling, flags.batch_size, flags.device, depth=flags.ring_depth) \
        if flags.preprocess == "hip" else None
    tensors = None if stage else [host_preprocess(p, spec, flags.scaling) for p in pixels]

    done = queue.Queue()
    responses, pending, sent_names = [], [], []
    callbacks_outstanding = 0
    slot_of = {}  # request id -> ring slot its input occupies

    def request_id(r):
        raw = r.get_response()
        return raw["id"] if isinstance(raw, dict) else raw.id

    def collect_one():
        """Waits for one outstanding request (the oldest future, or whichever callback fires next)
        and frees the ring slot its input used."""
        global callbacks_outstanding
        if pending:
            result = pending.pop(0).get_result()
        else:
            result, error = done.get(timeout=60)
            callbacks_outstanding -= 1
            if error is not None:
                fail("inference failed: " + str(error))
        responses.append(result)
        slot = slot_of.pop(request_id(result), None)
        if slot is not None:
            stage.release(slot)

    def drain():
        while pending or callbacks_outstanding:
            collect_one()

    if flags.streaming:
        client.start_stream(partial(completion, done))
    try:
        dims = [spec.c, spec.h, spec.w] if spec.layout == "NCHW" else [spec.h, spec.w, spec.c]
        for sent, batch in enumerate(batches(filenames, flags.batch_size), start=1):
            inp = mod.InferInput(spec.input_name, ([len(batch)] if batched else []) + dims, spec.dtype)
            if stage:
                while not stage.has_free_slot():  # ring full: the oldest request still reads its slot
                    collect_one()
                region, nbytes, offset, slot = stage.fill([pixels[i] for i in batch])
                inp.set_shared_memory(region, nbytes, offset=offset)
                slot_of[str(sent)] = slot
            else:
                inp.set_data_from_numpy(np.stack([tensors[i] for i in batch]) if batched else tensors[batch[0]])
            out = mod.InferRequestedOutput(spec.output_name, class_count=flags.classes)
            sent_names.append([filenames[i] for i in batch])
            kwargs = dict(request_id=str(sent), model_version=flags.model_version, outputs=[out])
            if flags.streaming:
                client.async_stream_infer(flags.model_name, [inp], **kwargs)
                callbacks_outstanding += 1
            elif flags.async_set and grpc_mode:
                client.async_infer(flags.model_name, [inp], partial(completion, done), **kwargs)
                callbacks_outstanding += 1
            elif flags.async_set:
                pending.append(client.async_infer(flags.model_name, [inp], **kwargs))
            else:
                responses.append(client.infer(flags.model_name, [inp], **kwargs))
                if stage:
                    stage.release(slot_of.pop(str(sent)))
        drain()
    except InferenceServerException as e:
        fail("inference failed: " + str(e))
    finally:
        if flags.streaming:
            client.stop_stream()

    for response in sorted(responses, key=lambda r: int(request_id(r))):
        rid = request_id(response)
        print_classes(response, spec.output_name, flags.batch_size, batched, rid, sent_names[int(rid) - 1])
    if stage:
        stage.close()
    client.close()
    print("PASS")
