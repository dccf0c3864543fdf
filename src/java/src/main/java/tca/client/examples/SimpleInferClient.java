package tca.client.examples;

import java.util.List;
import tca.client.InferInput;
import tca.client.InferRequestedOutput;
import tca.client.InferResult;
import tca.client.InferenceServerClient;
import tca.client.endpoint.FixedEndpoint;
import tca.client.pojo.DataType;

/**
 * One request to the {@code simple} add/sub model: INPUT0 as binary data, INPUT1 inside the JSON header,
 * OUTPUT0 binary, OUTPUT1 JSON - the four delivery combinations of the protocol in one exchange.
 * Usage: {@code SimpleInferClient [host:port]}.
 */
public final class SimpleInferClient {
  private SimpleInferClient() {}

  public static void main(String[] args) throws Exception {
    String url = args.length > 0 ? args[0] : "localhost:8000";
    int[] a = new int[16];
    int[] b = new int[16];
    for (int i = 0; i < 16; i++) {
      a[i] = i;
      b[i] = 1;
    }
    InferInput in0 = new InferInput("INPUT0", new long[] {1, 16}, DataType.INT32);
    in0.setData(a, true);
    InferInput in1 = new InferInput("INPUT1", new long[] {1, 16}, DataType.INT32);
    in1.setData(b, false);
    List<InferRequestedOutput> outputs =
        List.of(new InferRequestedOutput("OUTPUT0", true), new InferRequestedOutput("OUTPUT1", false));
    try (InferenceServerClient client = new InferenceServerClient(new FixedEndpoint(url), 5000, 5000)) {
      if (!client.isServerLive() || !client.isModelReady("simple")) {
        System.err.println("server or model not ready at " + url);
        System.exit(1);
      }
      InferResult result = client.infer("simple", List.of(in0, in1), outputs);
      int[] sum = result.getOutputAsInt("OUTPUT0");
      int[] diff = result.getOutputAsInt("OUTPUT1");
      for (int i = 0; i < 16; i++) {
        System.out.printf("%d + %d = %d, %d - %d = %d%n", a[i], b[i], sum[i], a[i], b[i], diff[i]);
        if (sum[i] != a[i] + b[i] || diff[i] != a[i] - b[i]) {
          System.err.println("incorrect result at " + i);
          System.exit(1);
        }
      }
    }
    System.out.println("PASS");
  }
}
