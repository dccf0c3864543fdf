package tca.client.examples;

import java.util.List;
import tca.client.InferInput;
import tca.client.InferRequestedOutput;
import tca.client.InferResult;
import tca.client.InferenceServerClient;
import tca.client.endpoint.FixedEndpoint;
import tca.client.pojo.DataType;

/**
 * Sends many requests to {@code custom_identity_int32}, creating a new client every 100 requests, and
 * reports heap use so that a leak in the client shows up as steady growth.
 * Usage: {@code MemoryGrowthTest [host:port] [requests]}.
 */
public final class MemoryGrowthTest {
  private MemoryGrowthTest() {}

  private static long usedHeap() {
    System.gc();
    Runtime rt = Runtime.getRuntime();
    return rt.totalMemory() - rt.freeMemory();
  }

  public static void main(String[] args) throws Exception {
    String url = args.length > 0 ? args[0] : "localhost:8000";
    int requests = args.length > 1 ? Integer.parseInt(args[1]) : 10000;
    int[] data = new int[16];
    for (int i = 0; i < 16; i++) {
      data[i] = i;
    }
    InferInput in = new InferInput("INPUT0", new long[] {1, 16}, DataType.INT32);
    in.setData(data, true);
    List<InferRequestedOutput> outputs = List.of(new InferRequestedOutput("OUTPUT0", true));
    long before = usedHeap();
    InferenceServerClient client = new InferenceServerClient(new FixedEndpoint(url), 5000, 5000);
    try {
      for (int i = 0; i < requests; i++) {
        if (i > 0 && i % 100 == 0) {
          client.close();
          client = new InferenceServerClient(new FixedEndpoint(url), 5000, 5000);
        }
        InferResult r = client.infer("custom_identity_int32", List.of(in), outputs);
        int[] out = r.getOutputAsInt("OUTPUT0");
        if (out.length != 16 || out[15] != 15) {
          System.err.println("identity output differs from its input");
          System.exit(1);
        }
        if ((i + 1) % 1000 == 0) {
          System.out.printf("%d requests, heap in use: %d KiB%n", i + 1, usedHeap() / 1024);
        }
      }
    } finally {
      client.close();
    }
    long after = usedHeap();
    System.out.printf("heap before %d KiB, after %d KiB%n", before / 1024, after / 1024);
    System.out.println("PASS");
  }
}
