package clients;

import com.google.protobuf.ByteString;
import inference.GRPCInferenceServiceGrpc;
import inference.GrpcService.ModelInferRequest;
import inference.GrpcService.ModelInferResponse;
import inference.GrpcService.ModelMetadataRequest;
import inference.GrpcService.ServerLiveRequest;
import inference.GrpcService.ServerReadyRequest;
import io.grpc.ManagedChannel;
import io.grpc.ManagedChannelBuilder;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.IntBuffer;

/**
 * Java client on generated stubs: health checks, model metadata and one inference on the "simple"
 * add/sub model with little-endian tensors in raw_input_contents.
 * Usage: {@code mvn exec:java -Dexec.args="localhost 8001"}.
 */
public final class SimpleJavaClient {
  private static final int ELEMENTS = 16;

  private SimpleJavaClient() {}

  private static ByteString pack(int[] values) {
    ByteBuffer buffer = ByteBuffer.allocate(values.length * Integer.BYTES).order(ByteOrder.LITTLE_ENDIAN);
    buffer.asIntBuffer().put(values);
    return ByteString.copyFrom(buffer.array());
  }

  private static int[] unpack(ByteString raw) {
    IntBuffer view = raw.asReadOnlyByteBuffer().order(ByteOrder.LITTLE_ENDIAN).asIntBuffer();
    int[] out = new int[view.remaining()];
    view.get(out);
    return out;
  }

  public static void main(String[] args) {
    String host = args.length > 0 ? args[0] : "localhost";
    int port = args.length > 1 ? Integer.parseInt(args[1]) : 8001;
    ManagedChannel channel = ManagedChannelBuilder.forAddress(host, port).usePlaintext().build();
    try {
      GRPCInferenceServiceGrpc.GRPCInferenceServiceBlockingStub stub = GRPCInferenceServiceGrpc.newBlockingStub(channel);
      if (!stub.serverLive(ServerLiveRequest.getDefaultInstance()).getLive()
          || !stub.serverReady(ServerReadyRequest.getDefaultInstance()).getReady()) {
        throw new IllegalStateException("server is not live/ready");
      }
      System.out.println("model: " + stub.modelMetadata(ModelMetadataRequest.newBuilder().setName("simple").build()).getName());

      int[] in0 = new int[ELEMENTS];
      int[] in1 = new int[ELEMENTS];
      for (int i = 0; i < ELEMENTS; i++) {
        in0[i] = i;
        in1[i] = 1;
      }
      ModelInferRequest.Builder request = ModelInferRequest.newBuilder().setModelName("simple");
      for (String name : new String[] {"INPUT0", "INPUT1"}) {
        request.addInputs(ModelInferRequest.InferInputTensor.newBuilder()
            .setName(name).setDatatype("INT32").addShape(1).addShape(ELEMENTS));
      }
      request.addOutputs(ModelInferRequest.InferRequestedOutputTensor.newBuilder().setName("OUTPUT0"));
      request.addOutputs(ModelInferRequest.InferRequestedOutputTensor.newBuilder().setName("OUTPUT1"));
      request.addRawInputContents(pack(in0)).addRawInputContents(pack(in1));

      ModelInferResponse response = stub.modelInfer(request.build());
      int[] sum = unpack(response.getRawOutputContents(0));
      int[] diff = unpack(response.getRawOutputContents(1));
      for (int i = 0; i < ELEMENTS; i++) {
        System.out.printf("%d + %d = %d    %d - %d = %d%n", in0[i], in1[i], sum[i], in0[i], in1[i], diff[i]);
        if (sum[i] != in0[i] + in1[i] || diff[i] != in0[i] - in1[i]) {
          throw new IllegalStateException("incorrect result at " + i);
        }
      }
      System.out.println("PASS");
    } finally {
      channel.shutdownNow();
    }
  }
}
