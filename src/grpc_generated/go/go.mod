module tca/grpc_simple_client

go 1.20

require (
	google.golang.org/grpc v1.58.0
	google.golang.org/protobuf v1.31.0
)
