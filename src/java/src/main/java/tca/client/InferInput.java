package tca.client;

import java.util.ArrayList;
import java.util.LinkedHashMap;
import java.util.List;
import java.util.Map;
import tca.client.pojo.DataType;
import tca.client.pojo.Parameters;

/**
 * One input tensor of a request.  Data is either carried as binary after the JSON header
 * ({@code binary = true}), embedded in the JSON header, or referenced in a registered shared-memory region.
 */
public class InferInput {
  private final String name;
  private long[] shape;
  private final DataType datatype;
  private final Parameters parameters = new Parameters();
  private byte[] binaryData;
  private List<Object> jsonData;

  public InferInput(String name, long[] shape, DataType datatype) {
    this.name = name;
    this.shape = shape.clone();
    this.datatype = datatype;
  }

  public String getName() {
    return name;
  }

  public DataType getDatatype() {
    return datatype;
  }

  public long[] getShape() {
    return shape;
  }

  public void setShape(long[] shape) {
    this.shape = shape.clone();
  }

  byte[] getBinaryData() {
    return binaryData;
  }

  private void check(DataType expected, int count) {
    if (datatype != expected) {
      throw new IllegalArgumentException("input '" + name + "' is " + datatype + ", data is " + expected);
    }
    if (count != Util.elementCount(shape)) {
      throw new IllegalArgumentException(
          "input '" + name + "': got " + count + " elements, the shape holds " + Util.elementCount(shape));
    }
  }

  private void store(byte[] bytes, List<Object> json, boolean binary) {
    parameters.remove(Parameters.KEY_SHM_REGION);
    parameters.remove(Parameters.KEY_SHM_BYTE_SIZE);
    parameters.remove(Parameters.KEY_SHM_OFFSET);
    if (binary) {
      binaryData = bytes;
      jsonData = null;
      parameters.put(Parameters.KEY_BINARY_DATA_SIZE, (long) bytes.length);
    } else {
      binaryData = null;
      jsonData = json;
      parameters.remove(Parameters.KEY_BINARY_DATA_SIZE);
    }
  }

  public void setData(boolean[] v, boolean binary) {
    check(DataType.BOOL, v.length);
    List<Object> json = new ArrayList<>(v.length);
    for (boolean b : v) {
      json.add(b);
    }
    store(binary ? BinaryProtocol.toBytes(v) : null, json, binary);
  }

  public void setData(byte[] v, boolean binary) {
    if (datatype != DataType.INT8 && datatype != DataType.UINT8) {
      throw new IllegalArgumentException("byte[] data needs an INT8 or UINT8 input");
    }
    check(datatype, v.length);
    List<Object> json = new ArrayList<>(v.length);
    for (byte b : v) {
      json.add(datatype == DataType.UINT8 ? (long) (b & 0xFF) : (long) b);
    }
    store(binary ? BinaryProtocol.toBytes(v) : null, json, binary);
  }

  public void setData(short[] v, boolean binary) {
    if (datatype != DataType.INT16 && datatype != DataType.UINT16) {
      throw new IllegalArgumentException("short[] data needs an INT16 or UINT16 input");
    }
    check(datatype, v.length);
    List<Object> json = new ArrayList<>(v.length);
    for (short s : v) {
      json.add(datatype == DataType.UINT16 ? (long) (s & 0xFFFF) : (long) s);
    }
    store(binary ? BinaryProtocol.toBytes(v) : null, json, binary);
  }

  public void setData(int[] v, boolean binary) {
    if (datatype != DataType.INT32 && datatype != DataType.UINT32) {
      throw new IllegalArgumentException("int[] data needs an INT32 or UINT32 input");
    }
    check(datatype, v.length);
    List<Object> json = new ArrayList<>(v.length);
    for (int i : v) {
      json.add(datatype == DataType.UINT32 ? (i & 0xFFFFFFFFL) : (long) i);
    }
    store(binary ? BinaryProtocol.toBytes(v) : null, json, binary);
  }

  public void setData(long[] v, boolean binary) {
    if (datatype != DataType.INT64 && datatype != DataType.UINT64) {
      throw new IllegalArgumentException("long[] data needs an INT64 or UINT64 input");
    }
    check(datatype, v.length);
    List<Object> json = new ArrayList<>(v.length);
    for (long l : v) {
      json.add(l);
    }
    store(binary ? BinaryProtocol.toBytes(v) : null, json, binary);
  }

  public void setData(float[] v, boolean binary) {
    if (datatype == DataType.BF16) {
      check(DataType.BF16, v.length);
      if (!binary) {
        throw new IllegalArgumentException("BF16 tensors can only be sent as binary data");
      }
      store(BinaryProtocol.toBf16Bytes(v), null, true);
      return;
    }
    check(DataType.FP32, v.length);
    List<Object> json = new ArrayList<>(v.length);
    for (float f : v) {
      json.add((double) f);
    }
    store(binary ? BinaryProtocol.toBytes(v) : null, json, binary);
  }

  public void setData(double[] v, boolean binary) {
    check(DataType.FP64, v.length);
    List<Object> json = new ArrayList<>(v.length);
    for (double d : v) {
      json.add(d);
    }
    store(binary ? BinaryProtocol.toBytes(v) : null, json, binary);
  }

  public void setData(String[] v, boolean binary) {
    check(DataType.BYTES, v.length);
    List<Object> json = new ArrayList<>(v.length);
    for (String s : v) {
      json.add(s);
    }
    store(binary ? BinaryProtocol.toBytes(v) : null, json, binary);
  }

  /** Already encoded tensor bytes (e.g. FP16 produced elsewhere). */
  public void setRawData(byte[] encoded) {
    store(encoded.clone(), null, true);
  }

  /** The tensor lives in a shared-memory region registered with the server (system or device memory). */
  public void setSharedMemory(String regionName, long byteSize, long offset) {
    binaryData = null;
    jsonData = null;
    parameters.remove(Parameters.KEY_BINARY_DATA_SIZE);
    parameters.put(Parameters.KEY_SHM_REGION, regionName);
    parameters.put(Parameters.KEY_SHM_BYTE_SIZE, byteSize);
    if (offset != 0) {
      parameters.put(Parameters.KEY_SHM_OFFSET, offset);
    }
  }

  Map<String, Object> toJson() {
    Map<String, Object> m = new LinkedHashMap<>();
    m.put("name", name);
    m.put("shape", shape);
    m.put("datatype", datatype.name());
    if (!parameters.isEmpty()) {
      m.put("parameters", parameters.asMap());
    }
    if (jsonData != null) {
      m.put("data", jsonData);
    }
    return m;
  }
}
