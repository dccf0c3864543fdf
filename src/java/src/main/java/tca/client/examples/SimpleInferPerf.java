package tca.client.examples;

import java.util.ArrayList;
import java.util.Arrays;
import java.util.List;
import java.util.concurrent.atomic.DoubleAdder;
import tca.client.InferInput;
import tca.client.InferRequestedOutput;
import tca.client.InferenceServerClient;
import tca.client.InferenceServerClient.InferArguments;
import tca.client.endpoint.AbstractEndpoint;
import tca.client.endpoint.FixedEndpoint;
import tca.client.endpoint.RoundRobinEndpoint;
import tca.client.pojo.DataType;

/**
 * Closed-loop throughput probe: T threads, each with its own client, send N synchronous requests to
 * {@code simple}; prints the running latency every 100 requests, per-thread QPS and the totals.
 * Usage: {@code SimpleInferPerf [host:port[,host:port...]] [threads] [requests]} - several addresses
 * (one server per GPU) are driven round-robin.
 */
public final class SimpleInferPerf {
  private SimpleInferPerf() {}

  public static void main(String[] args) throws Exception {
    final String[] urls = (args.length > 0 ? args[0] : "localhost:8000").split(",");
    final int threads = args.length > 1 ? Integer.parseInt(args[1]) : 8;
    final int requests = args.length > 2 ? Integer.parseInt(args[2]) : 1000;
    final int gap = 100;

    int[] data0 = new int[16];
    int[] data1 = new int[16];
    for (int i = 0; i < 16; i++) {
      data0[i] = i + 1;
      data1[i] = i + 1;
    }
    InferInput in0 = new InferInput("INPUT0", new long[] {1, 16}, DataType.INT32);
    in0.setData(data0, true);
    InferInput in1 = new InferInput("INPUT1", new long[] {1, 16}, DataType.INT32);
    in1.setData(data1, true);
    final List<InferInput> inputs = List.of(in0, in1);
    final List<InferRequestedOutput> outputs =
        List.of(new InferRequestedOutput("OUTPUT0", true), new InferRequestedOutput("OUTPUT1", true));

    System.out.printf("Testing simple with %d threads x %d requests against %s%n", threads, requests, Arrays.toString(urls));
    final DoubleAdder totalQps = new DoubleAdder();
    final DoubleAdder totalLatency = new DoubleAdder();
    List<Thread> pool = new ArrayList<>();
    for (int t = 0; t < threads; t++) {
      Thread worker = new Thread(() -> {
        long tid = Thread.currentThread().getId();
        AbstractEndpoint endpoint =
            urls.length == 1 ? new FixedEndpoint(urls[0]) : new RoundRobinEndpoint(Arrays.asList(urls));
        try (InferenceServerClient client = new InferenceServerClient(endpoint, 5000, 5000)) {
          InferArguments call = new InferArguments("simple", inputs, outputs);
          long start = System.nanoTime();
          long gapStart = start;
          for (int i = 0; i < requests; i++) {
            client.infer(call);
            if ((i + 1) % gap == 0) {
              long now = System.nanoTime();
              System.out.printf("[%d][GAP] requests: %d, avg latency(ms): %.3f%n", tid, i + 1, (now - gapStart) / 1e6 / gap);
              gapStart = now;
            }
          }
          double seconds = (System.nanoTime() - start) / 1e9;
          double qps = requests / seconds;
          double latencyMs = 1000.0 * seconds / requests;
          System.out.printf("[%d][TOTAL] avg latency(ms): %.3f, qps: %.1f%n", tid, latencyMs, qps);
          totalQps.add(qps);
          totalLatency.add(latencyMs);
        } catch (Exception e) {
          e.printStackTrace();
        }
      });
      worker.start();
      pool.add(worker);
    }
    for (Thread w : pool) {
      w.join();
    }
    System.out.println("==================================");
    System.out.printf("[ALL]         QPS: %.1f%n", totalQps.sum());
    System.out.printf("[ALL] latency(ms): %.3f%n", totalLatency.sum() / threads);
    System.out.println("==================================");
  }
}
