package tca.client;

/** Failure of an inference request: transport error, HTTP error status or a protocol violation. */
public class InferenceException extends Exception {
  private static final long serialVersionUID = 1L;
  private final int statusCode;

  public InferenceException(String message) {
    this(message, -1, null);
  }

  public InferenceException(String message, int statusCode) {
    this(message, statusCode, null);
  }

  public InferenceException(String message, int statusCode, Throwable cause) {
    super(message, cause);
    this.statusCode = statusCode;
  }

  /** HTTP status of the failed exchange, or -1 when the request never produced one. */
  public int getStatusCode() {
    return statusCode;
  }
}
