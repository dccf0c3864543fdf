package tca.client;

import java.net.URI;
import java.net.http.HttpClient;
import java.net.http.HttpRequest;
import java.net.http.HttpResponse;
import java.nio.charset.StandardCharsets;
import java.time.Duration;
import java.util.ArrayList;
import java.util.LinkedHashMap;
import java.util.List;
import java.util.Map;
import java.util.concurrent.CompletableFuture;
import java.util.concurrent.ExecutorService;
import java.util.concurrent.Executors;
import tca.client.endpoint.AbstractEndpoint;
import tca.client.pojo.Parameters;

/**
 * HTTP client of the KServe-v2 / Triton inference protocol.
 *
 * <p>Requests are a JSON header followed by the binary tensors (header length announced in
 * {@code Inference-Header-Content-Length}).  A failed attempt is retried on the next endpoint of the
 * provider (one server per GPU is the usual multi-GPU deployment).  Besides the blocking
 * {@link #infer(InferArguments)} there is {@link #inferAsync(InferArguments)} on the JDK's
 * asynchronous HTTP client.
 */
public class InferenceServerClient implements AutoCloseable {
  public static final String HEADER_LENGTH = "Inference-Header-Content-Length";

  private final AbstractEndpoint endpoint;
  private final HttpClient http;
  private final ExecutorService executor;
  private final Duration requestTimeout;
  private int retryCount = 3;

  /**
   * @param endpoint address provider
   * @param connectTimeoutMs TCP connect timeout
   * @param requestTimeoutMs timeout of one HTTP exchange
   */
  public InferenceServerClient(AbstractEndpoint endpoint, int connectTimeoutMs, int requestTimeoutMs) {
    this.endpoint = endpoint;
    this.executor = Executors.newFixedThreadPool(Math.max(4, Runtime.getRuntime().availableProcessors() / 2), r -> {
      Thread t = new Thread(r, "tca-http");
      t.setDaemon(true);
      return t;
    });
    this.http = HttpClient.newBuilder()
        .version(HttpClient.Version.HTTP_1_1)
        .connectTimeout(Duration.ofMillis(connectTimeoutMs))
        .executor(executor)
        .build();
    this.requestTimeout = Duration.ofMillis(requestTimeoutMs);
  }

  public void setRetryCount(int retryCount) {
    this.retryCount = Math.max(1, retryCount);
  }

  /** Everything that describes one inference request. */
  public static class InferArguments {
    final String modelName;
    String modelVersion = "";
    final List<InferInput> inputs;
    final List<InferRequestedOutput> outputs;
    String requestId = "";
    long sequenceId = 0;
    boolean sequenceStart = false;
    boolean sequenceEnd = false;
    long priority = 0;
    long timeoutMicros = 0;
    final Map<String, String> headers = new LinkedHashMap<>();
    final Map<String, String> queryParams = new LinkedHashMap<>();
    final Parameters custom = new Parameters();

    public InferArguments(String modelName, List<InferInput> inputs, List<InferRequestedOutput> outputs) {
      this.modelName = modelName;
      this.inputs = inputs;
      this.outputs = outputs == null ? new ArrayList<>() : outputs;
    }

    public InferArguments modelVersion(String v) {
      this.modelVersion = v;
      return this;
    }

    public InferArguments requestId(String v) {
      this.requestId = v;
      return this;
    }

    public InferArguments sequence(long id, boolean start, boolean end) {
      this.sequenceId = id;
      this.sequenceStart = start;
      this.sequenceEnd = end;
      return this;
    }

    public InferArguments priority(long v) {
      this.priority = v;
      return this;
    }

    public InferArguments timeoutMicros(long v) {
      this.timeoutMicros = v;
      return this;
    }

    public InferArguments header(String k, String v) {
      headers.put(k, v);
      return this;
    }

    public InferArguments queryParam(String k, String v) {
      queryParams.put(k, v);
      return this;
    }

    public InferArguments parameter(String k, Object v) {
      custom.put(k, v);
      return this;
    }
  }

  /** Serialised request: JSON header length and the whole body. */
  static final class Encoded {
    final int headerLength;
    final byte[] body;

    Encoded(int headerLength, byte[] body) {
      this.headerLength = headerLength;
      this.body = body;
    }
  }

  static Encoded encode(InferArguments a) {
    Map<String, Object> doc = new LinkedHashMap<>();
    if (!Util.isEmpty(a.requestId)) {
      doc.put("id", a.requestId);
    }
    Parameters p = new Parameters(a.custom.asMap());
    if (a.sequenceId != 0) {
      p.put(Parameters.KEY_SEQUENCE_ID, a.sequenceId);
      p.put(Parameters.KEY_SEQUENCE_START, a.sequenceStart);
      p.put(Parameters.KEY_SEQUENCE_END, a.sequenceEnd);
    }
    if (a.priority != 0) {
      p.put(Parameters.KEY_PRIORITY, a.priority);
    }
    if (a.timeoutMicros != 0) {
      p.put(Parameters.KEY_TIMEOUT, a.timeoutMicros);
    }
    if (a.outputs.isEmpty()) {
      p.put(Parameters.KEY_BINARY_DATA_OUTPUT, true);  // no outputs named: ask for all of them, binary
    }
    if (!p.isEmpty()) {
      doc.put("parameters", p.asMap());
    }
    List<Object> inputs = new ArrayList<>();
    int binaryBytes = 0;
    for (InferInput in : a.inputs) {
      inputs.add(in.toJson());
      if (in.getBinaryData() != null) {
        binaryBytes += in.getBinaryData().length;
      }
    }
    doc.put("inputs", inputs);
    if (!a.outputs.isEmpty()) {
      List<Object> outputs = new ArrayList<>();
      for (InferRequestedOutput o : a.outputs) {
        outputs.add(o.toJson());
      }
      doc.put("outputs", outputs);
    }
    byte[] header = Json.write(doc).getBytes(StandardCharsets.UTF_8);
    byte[] body = new byte[header.length + binaryBytes];
    System.arraycopy(header, 0, body, 0, header.length);
    int cursor = header.length;
    for (InferInput in : a.inputs) {
      byte[] b = in.getBinaryData();
      if (b != null) {
        System.arraycopy(b, 0, body, cursor, b.length);
        cursor += b.length;
      }
    }
    return new Encoded(header.length, body);
  }

  private HttpRequest build(String address, InferArguments a, Encoded enc) {
    StringBuilder url = new StringBuilder("http://").append(address).append("/v2/models/").append(Util.urlEncode(a.modelName));
    if (!Util.isEmpty(a.modelVersion)) {
      url.append("/versions/").append(Util.urlEncode(a.modelVersion));
    }
    url.append("/infer");
    char sep = '?';
    for (Map.Entry<String, String> q : a.queryParams.entrySet()) {
      url.append(sep).append(Util.urlEncode(q.getKey())).append('=').append(Util.urlEncode(q.getValue()));
      sep = '&';
    }
    HttpRequest.Builder b = HttpRequest.newBuilder(URI.create(url.toString()))
        .timeout(requestTimeout)
        .header("Content-Type", "application/octet-stream")
        .header(HEADER_LENGTH, Integer.toString(enc.headerLength))
        .POST(HttpRequest.BodyPublishers.ofByteArray(enc.body));
    for (Map.Entry<String, String> h : a.headers.entrySet()) {
      b.header(h.getKey(), h.getValue());
    }
    return b.build();
  }

  private static InferResult parse(HttpResponse<byte[]> resp) throws InferenceException {
    if (resp.statusCode() != 200) {
      String message = new String(resp.body(), StandardCharsets.UTF_8);
      try {
        Object parsed = Json.parse(message);
        if (parsed instanceof Map && ((Map<?, ?>) parsed).get("error") != null) {
          message = String.valueOf(((Map<?, ?>) parsed).get("error"));
        }
      } catch (InferenceException ignored) {
        // not JSON: report the body as it is
      }
      throw new InferenceException(message, resp.statusCode());
    }
    int headerLength = resp.headers().firstValue(HEADER_LENGTH).map(Integer::parseInt).orElse(0);
    return new InferResult(resp.body(), headerLength);
  }

  /** Blocking inference with retries over the endpoint provider. */
  public InferResult infer(InferArguments args) throws InferenceException {
    Encoded enc = encode(args);
    String failed = null;
    InferenceException last = null;
    for (int attempt = 0; attempt < retryCount; attempt++) {
      String address;
      try {
        address = endpoint.next(failed);
      } catch (Exception e) {
        throw new InferenceException("no endpoint available: " + e.getMessage(), -1, e);
      }
      try {
        HttpResponse<byte[]> resp = http.send(build(address, args, enc), HttpResponse.BodyHandlers.ofByteArray());
        InferResult result = parse(resp);
        endpoint.report(address, true);
        return result;
      } catch (InferenceException e) {
        if (e.getStatusCode() >= 400 && e.getStatusCode() < 500) {
          throw e;  // the request itself is wrong: another server will not like it better
        }
        last = e;
      } catch (InterruptedException e) {
        Thread.currentThread().interrupt();
        throw new InferenceException("interrupted", -1, e);
      } catch (Exception e) {
        last = new InferenceException("request to " + address + " failed: " + e, -1, e);
      }
      endpoint.report(address, false);
      failed = address;
    }
    throw last != null ? last : new InferenceException("inference failed");
  }

  public InferResult infer(String modelName, List<InferInput> inputs, List<InferRequestedOutput> outputs)
      throws InferenceException {
    return infer(new InferArguments(modelName, inputs, outputs));
  }

  /** Non-blocking inference (single attempt); the future fails with {@link InferenceException}. */
  public CompletableFuture<InferResult> inferAsync(InferArguments args) {
    final Encoded enc = encode(args);
    final String address;
    try {
      address = endpoint.next(null);
    } catch (Exception e) {
      return CompletableFuture.failedFuture(new InferenceException("no endpoint available: " + e.getMessage(), -1, e));
    }
    return http.sendAsync(build(address, args, enc), HttpResponse.BodyHandlers.ofByteArray()).handle((resp, err) -> {
      if (err != null) {
        endpoint.report(address, false);
        throw new java.util.concurrent.CompletionException(
            new InferenceException("request to " + address + " failed: " + err, -1, err));
      }
      try {
        InferResult r = parse(resp);
        endpoint.report(address, true);
        return r;
      } catch (InferenceException e) {
        throw new java.util.concurrent.CompletionException(e);
      }
    });
  }

  private String get(String path) throws InferenceException {
    try {
      String address = endpoint.next(null);
      HttpRequest req = HttpRequest.newBuilder(URI.create("http://" + address + path)).timeout(requestTimeout).GET().build();
      HttpResponse<byte[]> resp = http.send(req, HttpResponse.BodyHandlers.ofByteArray());
      String body = new String(resp.body(), StandardCharsets.UTF_8);
      if (resp.statusCode() != 200) {
        throw new InferenceException(body, resp.statusCode());
      }
      return body;
    } catch (InferenceException e) {
      throw e;
    } catch (InterruptedException e) {
      Thread.currentThread().interrupt();
      throw new InferenceException("interrupted", -1, e);
    } catch (Exception e) {
      throw new InferenceException("request failed: " + e, -1, e);
    }
  }

  public boolean isServerLive() {
    try {
      get("/v2/health/live");
      return true;
    } catch (InferenceException e) {
      return false;
    }
  }

  public boolean isServerReady() {
    try {
      get("/v2/health/ready");
      return true;
    } catch (InferenceException e) {
      return false;
    }
  }

  public boolean isModelReady(String modelName) {
    try {
      get("/v2/models/" + Util.urlEncode(modelName) + "/ready");
      return true;
    } catch (InferenceException e) {
      return false;
    }
  }

  /** Model metadata as parsed JSON ({@code Map<String,Object>}). */
  public Object getModelMetadata(String modelName) throws InferenceException {
    return Json.parse(get("/v2/models/" + Util.urlEncode(modelName)));
  }

  public Object getModelConfig(String modelName) throws InferenceException {
    return Json.parse(get("/v2/models/" + Util.urlEncode(modelName) + "/config"));
  }

  @Override
  public void close() {
    executor.shutdownNow();
  }
}
