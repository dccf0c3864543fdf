// Go client on generated stubs (run ./gen_go_stubs.sh first): health checks, model metadata and one
// inference on the "simple" add/sub model with tensors carried in raw_input_contents.
//
//	go run grpc_simple_client.go -u localhost:8001
package main

import (
	"bytes"
	"context"
	"encoding/binary"
	"flag"
	"fmt"
	"log"
	"time"

	"google.golang.org/grpc"
	"google.golang.org/grpc/credentials/insecure"

	pb "tca/grpc_simple_client/grpc-client"
)

const (
	elements  = 16
	modelName = "simple"
)

// int32 slice -> little-endian bytes, the wire form of every fixed-size datatype.
func pack(values []int32) []byte {
	buf := new(bytes.Buffer)
	for _, v := range values {
		if err := binary.Write(buf, binary.LittleEndian, v); err != nil {
			log.Fatalf("pack: %v", err)
		}
	}
	return buf.Bytes()
}

func unpack(raw []byte) []int32 {
	out := make([]int32, len(raw)/4)
	if err := binary.Read(bytes.NewReader(raw), binary.LittleEndian, &out); err != nil {
		log.Fatalf("unpack: %v", err)
	}
	return out
}

func main() {
	url := flag.String("u", "localhost:8001", "inference server gRPC address")
	version := flag.String("x", "", "model version (latest when empty)")
	flag.Parse()

	conn, err := grpc.Dial(*url, grpc.WithTransportCredentials(insecure.NewCredentials()))
	if err != nil {
		log.Fatalf("cannot connect to %s: %v", *url, err)
	}
	defer conn.Close()
	client := pb.NewGRPCInferenceServiceClient(conn)
	ctx, cancel := context.WithTimeout(context.Background(), 10*time.Second)
	defer cancel()

	live, err := client.ServerLive(ctx, &pb.ServerLiveRequest{})
	if err != nil || !live.Live {
		log.Fatalf("server is not live: %v", err)
	}
	ready, err := client.ServerReady(ctx, &pb.ServerReadyRequest{})
	if err != nil || !ready.Ready {
		log.Fatalf("server is not ready: %v", err)
	}
	meta, err := client.ModelMetadata(ctx, &pb.ModelMetadataRequest{Name: modelName, Version: *version})
	if err != nil {
		log.Fatalf("model metadata: %v", err)
	}
	fmt.Printf("model %q: %d inputs, %d outputs\n", meta.Name, len(meta.Inputs), len(meta.Outputs))

	in0 := make([]int32, elements)
	in1 := make([]int32, elements)
	for i := range in0 {
		in0[i] = int32(i)
		in1[i] = 1
	}
	request := &pb.ModelInferRequest{
		ModelName:    modelName,
		ModelVersion: *version,
		Inputs: []*pb.ModelInferRequest_InferInputTensor{
			{Name: "INPUT0", Datatype: "INT32", Shape: []int64{1, elements}},
			{Name: "INPUT1", Datatype: "INT32", Shape: []int64{1, elements}},
		},
		Outputs: []*pb.ModelInferRequest_InferRequestedOutputTensor{
			{Name: "OUTPUT0"}, {Name: "OUTPUT1"},
		},
		RawInputContents: [][]byte{pack(in0), pack(in1)},
	}
	response, err := client.ModelInfer(ctx, request)
	if err != nil {
		log.Fatalf("inference failed: %v", err)
	}
	if len(response.RawOutputContents) != 2 {
		log.Fatalf("expected 2 raw outputs, got %d", len(response.RawOutputContents))
	}
	sum := unpack(response.RawOutputContents[0])
	diff := unpack(response.RawOutputContents[1])
	for i := 0; i < elements; i++ {
		fmt.Printf("%d + %d = %d    %d - %d = %d\n", in0[i], in1[i], sum[i], in0[i], in1[i], diff[i])
		if sum[i] != in0[i]+in1[i] || diff[i] != in0[i]-in1[i] {
			log.Fatalf("incorrect result at %d", i)
		}
	}
	fmt.Println("PASS")
}
