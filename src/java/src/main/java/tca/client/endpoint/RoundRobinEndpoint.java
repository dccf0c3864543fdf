package tca.client.endpoint;

import java.util.ArrayList;
import java.util.List;
import java.util.concurrent.ConcurrentHashMap;
import java.util.concurrent.atomic.AtomicLong;

/**
 * Several servers, one per GPU of a node (or per node): requests rotate over them and an endpoint that
 * keeps failing is skipped for a cool-down period.
 */
public class RoundRobinEndpoint extends AbstractEndpoint {
  private final List<String> addresses = new ArrayList<>();
  private final AtomicLong cursor = new AtomicLong();
  private final ConcurrentHashMap<String, Long> quarantinedUntil = new ConcurrentHashMap<>();
  private final long coolDownMillis;

  public RoundRobinEndpoint(List<String> addresses) {
    this(addresses, 2000);
  }

  public RoundRobinEndpoint(List<String> addresses, long coolDownMillis) {
    if (addresses == null || addresses.isEmpty()) {
      throw new IllegalArgumentException("at least one endpoint is required");
    }
    for (String a : addresses) {
      this.addresses.add(new FixedEndpoint(a).pick());
    }
    this.coolDownMillis = coolDownMillis;
  }

  @Override
  protected String pick() {
    final long now = System.currentTimeMillis();
    for (int i = 0; i < addresses.size(); i++) {
      String a = addresses.get((int) (cursor.getAndIncrement() % addresses.size()));
      Long until = quarantinedUntil.get(a);
      if (until == null || until <= now) {
        return a;
      }
    }
    // everything is quarantined: try the next one anyway
    return addresses.get((int) (cursor.getAndIncrement() % addresses.size()));
  }

  @Override
  public int size() {
    return addresses.size();
  }

  @Override
  public void report(String endpoint, boolean ok) {
    if (ok) {
      quarantinedUntil.remove(endpoint);
    } else {
      quarantinedUntil.put(endpoint, System.currentTimeMillis() + coolDownMillis);
    }
  }
}
