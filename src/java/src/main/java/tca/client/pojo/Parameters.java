package tca.client.pojo;

import java.util.LinkedHashMap;
import java.util.Map;

/** The free-form "parameters" object of requests, tensors and responses (string / number / boolean values). */
public class Parameters {
  public static final String KEY_BINARY_DATA = "binary_data";
  public static final String KEY_BINARY_DATA_SIZE = "binary_data_size";
  public static final String KEY_BINARY_DATA_OUTPUT = "binary_data_output";
  public static final String KEY_CLASSIFICATION = "classification";
  public static final String KEY_SEQUENCE_ID = "sequence_id";
  public static final String KEY_SEQUENCE_START = "sequence_start";
  public static final String KEY_SEQUENCE_END = "sequence_end";
  public static final String KEY_PRIORITY = "priority";
  public static final String KEY_TIMEOUT = "timeout";
  public static final String KEY_SHM_REGION = "shared_memory_region";
  public static final String KEY_SHM_BYTE_SIZE = "shared_memory_byte_size";
  public static final String KEY_SHM_OFFSET = "shared_memory_offset";

  private final Map<String, Object> values = new LinkedHashMap<>();

  public Parameters() {}

  public Parameters(Map<String, Object> initial) {
    if (initial != null) {
      values.putAll(initial);
    }
  }

  public Parameters put(String key, Object value) {
    if (!(value instanceof String) && !(value instanceof Number) && !(value instanceof Boolean)) {
      throw new IllegalArgumentException("parameter '" + key + "' must be a string, a number or a boolean");
    }
    values.put(key, value);
    return this;
  }

  public Object get(String key) {
    return values.get(key);
  }

  public boolean has(String key) {
    return values.containsKey(key);
  }

  public Object remove(String key) {
    return values.remove(key);
  }

  public boolean getBool(String key, boolean dflt) {
    Object v = values.get(key);
    return v instanceof Boolean ? (Boolean) v : dflt;
  }

  public long getLong(String key, long dflt) {
    Object v = values.get(key);
    return v instanceof Number ? ((Number) v).longValue() : dflt;
  }

  public String getString(String key, String dflt) {
    Object v = values.get(key);
    return v instanceof String ? (String) v : dflt;
  }

  public boolean isEmpty() {
    return values.isEmpty();
  }

  public Map<String, Object> asMap() {
    return values;
  }
}
