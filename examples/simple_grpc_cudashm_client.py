#!/usr/bin/env python3
"""Device shared memory over GRPC.  The module is imported under its
historical name ``tritonclient.utils.cuda_shared_memory``; on this stack it is
hipshm: HIP IPC handles on an MI355X.  With ``--string`` the BYTES variant on
``simple_string`` runs instead.  (Reference:
``src/python/examples/simple_grpc_cudashm_client.py``.)"""
from _common import parser
from _shm_example import run_simple, run_string

import tritonclient.grpc as client_module

if __name__ == "__main__":
    ap = parser(__doc__, 8001)
    ap.add_argument("-g", "--device", type=int, default=0, help="GPU ordinal holding the regions")
    ap.add_argument("--string", action="store_true", help="run the BYTES (simple_string) variant")
    flags = ap.parse_args()
    client = client_module.InferenceServerClient(url=flags.url, verbose=flags.verbose)
    (run_string if flags.string else run_simple)(client_module, client, "device", flags.device)
    client.close()
    print("PASS: device shared memory")
