package tca.client.endpoint;

/**
 * Source of server addresses ("host:port[/base-path]").  The client asks for an endpoint per attempt;
 * {@link #next(String)} receives the endpoint that just failed so that a retry goes somewhere else
 * whenever the provider knows more than one server.
 */
public abstract class AbstractEndpoint {
  private static final int MAX_DRAWS = 16;

  /** Any endpoint the provider currently considers usable. */
  protected abstract String pick() throws Exception;

  /** Number of distinct endpoints behind this provider (1 for a fixed address). */
  public abstract int size();

  /** Endpoint for the next attempt, different from {@code failed} when there is an alternative. */
  public String next(String failed) throws Exception {
    String candidate = pick();
    if (failed == null || size() <= 1) {
      return candidate;
    }
    for (int i = 0; i < MAX_DRAWS && candidate.equals(failed); i++) {
      candidate = pick();
    }
    return candidate;
  }

  /** Feedback hook: the request sent to {@code endpoint} succeeded or failed. */
  public void report(String endpoint, boolean ok) {}
}
