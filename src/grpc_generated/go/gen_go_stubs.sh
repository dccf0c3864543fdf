#!/bin/bash
# Generates the Go stubs of inference.GRPCInferenceService into ./grpc-client from the .proto files
# shipped in this repository.  Needs protoc, protoc-gen-go and protoc-gen-go-grpc on PATH.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
PROTOS=$(cd "$HERE/../../../triton_client_amd/grpc/protos" && pwd)
OUT="$HERE/grpc-client"
PKG="tca/grpc_simple_client/grpc-client"
mkdir -p "$OUT"
protoc -I "$PROTOS" \
  --go_out="$OUT" --go_opt=paths=source_relative \
  --go_opt=Mgrpc_service.proto="$PKG" --go_opt=Mmodel_config.proto="$PKG" \
  --go-grpc_out="$OUT" --go-grpc_opt=paths=source_relative \
  --go-grpc_opt=Mgrpc_service.proto="$PKG" --go-grpc_opt=Mmodel_config.proto="$PKG" \
  "$PROTOS/grpc_service.proto" "$PROTOS/model_config.proto"
echo "stubs written to $OUT"
