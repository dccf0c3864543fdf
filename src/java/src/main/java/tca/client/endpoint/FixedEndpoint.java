package tca.client.endpoint;

/** A single, fixed server address. */
public class FixedEndpoint extends AbstractEndpoint {
  private final String address;

  public FixedEndpoint(String address) {
    if (address == null || address.isEmpty()) {
      throw new IllegalArgumentException("endpoint address must not be empty");
    }
    String a = address;
    if (a.startsWith("http://")) {
      a = a.substring("http://".length());
    }
    while (a.endsWith("/")) {
      a = a.substring(0, a.length() - 1);
    }
    this.address = a;
  }

  @Override
  protected String pick() {
    return address;
  }

  @Override
  public int size() {
    return 1;
  }
}
