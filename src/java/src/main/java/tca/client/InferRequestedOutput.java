package tca.client;

import java.util.LinkedHashMap;
import java.util.Map;
import tca.client.pojo.Parameters;

/** One requested output: binary or JSON delivery, optional top-K classification, optional shared memory. */
public class InferRequestedOutput {
  private final String name;
  private final Parameters parameters = new Parameters();

  public InferRequestedOutput(String name, boolean binary) {
    this(name, binary, 0);
  }

  public InferRequestedOutput(String name, boolean binary, int classCount) {
    this.name = name;
    parameters.put(Parameters.KEY_BINARY_DATA, binary);
    if (classCount > 0) {
      parameters.put(Parameters.KEY_CLASSIFICATION, (long) classCount);
    }
  }

  public String getName() {
    return name;
  }

  public void setSharedMemory(String regionName, long byteSize, long offset) {
    parameters.remove(Parameters.KEY_BINARY_DATA);
    parameters.put(Parameters.KEY_SHM_REGION, regionName);
    parameters.put(Parameters.KEY_SHM_BYTE_SIZE, byteSize);
    if (offset != 0) {
      parameters.put(Parameters.KEY_SHM_OFFSET, offset);
    }
  }

  public void unsetSharedMemory() {
    parameters.remove(Parameters.KEY_SHM_REGION);
    parameters.remove(Parameters.KEY_SHM_BYTE_SIZE);
    parameters.remove(Parameters.KEY_SHM_OFFSET);
    parameters.put(Parameters.KEY_BINARY_DATA, true);
  }

  Map<String, Object> toJson() {
    Map<String, Object> m = new LinkedHashMap<>();
    m.put("name", name);
    if (!parameters.isEmpty()) {
      m.put("parameters", parameters.asMap());
    }
    return m;
  }
}
