package tca.client;

import java.nio.ByteBuffer;
import java.nio.charset.StandardCharsets;
import java.util.HashMap;
import java.util.List;
import java.util.Map;
import tca.client.pojo.DataType;
import tca.client.pojo.IOTensor;
import tca.client.pojo.InferenceResponse;
import tca.client.pojo.Parameters;

/**
 * A parsed inference response.  The body is the JSON header (its length given by the
 * {@code Inference-Header-Content-Length} response header) followed by the binary outputs in header order.
 */
public class InferResult {
  private final InferenceResponse response;
  private final byte[] body;
  private final Map<String, int[]> binaryRange = new HashMap<>();  // name -> {offset, length}

  public InferResult(byte[] body, int headerLength) throws InferenceException {
    this.body = body;
    final int jsonLength = headerLength > 0 ? headerLength : body.length;
    if (jsonLength > body.length) {
      throw new InferenceException("header length " + jsonLength + " exceeds the body (" + body.length + " bytes)");
    }
    this.response = Util.responseFromJson(Json.parse(new String(body, 0, jsonLength, StandardCharsets.UTF_8)));
    int cursor = jsonLength;
    for (IOTensor t : response.getOutputs()) {
      Parameters p = t.getParameters();
      if (p.has(Parameters.KEY_BINARY_DATA_SIZE)) {
        int size = (int) p.getLong(Parameters.KEY_BINARY_DATA_SIZE, 0);
        if (cursor + size > body.length) {
          throw new InferenceException("binary data of output '" + t.getName() + "' is truncated");
        }
        binaryRange.put(t.getName(), new int[] {cursor, size});
        cursor += size;
      }
    }
  }

  public InferenceResponse getResponse() {
    return response;
  }

  public IOTensor getOutput(String name) {
    return response.getOutputByName(name);
  }

  private IOTensor require(String name, DataType... accepted) throws InferenceException {
    IOTensor t = getOutput(name);
    if (t == null) {
      throw new InferenceException("the response has no output named '" + name + "'");
    }
    DataType actual = DataType.fromString(t.getDatatype());
    for (DataType a : accepted) {
      if (a == actual) {
        return t;
      }
    }
    throw new InferenceException("output '" + name + "' has datatype " + actual);
  }

  private ByteBuffer binary(String name) {
    int[] r = binaryRange.get(name);
    return r == null ? null : BinaryProtocol.view(body, r[0], r[1]);
  }

  private List<Object> json(IOTensor t) throws InferenceException {
    if (t.getData() == null) {
      throw new InferenceException("output '" + t.getName() + "' carries no data (written to shared memory?)");
    }
    return t.getData();
  }

  public boolean[] getOutputAsBool(String name) throws InferenceException {
    IOTensor t = require(name, DataType.BOOL);
    ByteBuffer b = binary(name);
    if (b != null) {
      boolean[] out = new boolean[b.remaining()];
      for (int i = 0; i < out.length; i++) {
        out[i] = b.get() != 0;
      }
      return out;
    }
    List<Object> d = json(t);
    boolean[] out = new boolean[d.size()];
    for (int i = 0; i < out.length; i++) {
      out[i] = (Boolean) d.get(i);
    }
    return out;
  }

  public byte[] getOutputAsByte(String name) throws InferenceException {
    IOTensor t = require(name, DataType.INT8, DataType.UINT8);
    ByteBuffer b = binary(name);
    if (b != null) {
      byte[] out = new byte[b.remaining()];
      b.get(out);
      return out;
    }
    List<Object> d = json(t);
    byte[] out = new byte[d.size()];
    for (int i = 0; i < out.length; i++) {
      out[i] = ((Number) d.get(i)).byteValue();
    }
    return out;
  }

  public short[] getOutputAsShort(String name) throws InferenceException {
    IOTensor t = require(name, DataType.INT16, DataType.UINT16, DataType.FP16, DataType.BF16);
    ByteBuffer b = binary(name);
    if (b != null) {
      short[] out = new short[b.remaining() / 2];
      b.asShortBuffer().get(out);
      return out;
    }
    List<Object> d = json(t);
    short[] out = new short[d.size()];
    for (int i = 0; i < out.length; i++) {
      out[i] = ((Number) d.get(i)).shortValue();
    }
    return out;
  }

  public int[] getOutputAsInt(String name) throws InferenceException {
    IOTensor t = require(name, DataType.INT32, DataType.UINT32);
    ByteBuffer b = binary(name);
    if (b != null) {
      int[] out = new int[b.remaining() / 4];
      b.asIntBuffer().get(out);
      return out;
    }
    List<Object> d = json(t);
    int[] out = new int[d.size()];
    for (int i = 0; i < out.length; i++) {
      out[i] = ((Number) d.get(i)).intValue();
    }
    return out;
  }

  public long[] getOutputAsLong(String name) throws InferenceException {
    IOTensor t = require(name, DataType.INT64, DataType.UINT64);
    ByteBuffer b = binary(name);
    if (b != null) {
      long[] out = new long[b.remaining() / 8];
      b.asLongBuffer().get(out);
      return out;
    }
    List<Object> d = json(t);
    long[] out = new long[d.size()];
    for (int i = 0; i < out.length; i++) {
      out[i] = ((Number) d.get(i)).longValue();
    }
    return out;
  }

  public float[] getOutputAsFloat(String name) throws InferenceException {
    IOTensor t = require(name, DataType.FP32, DataType.BF16);
    ByteBuffer b = binary(name);
    if (b != null) {
      if (DataType.fromString(t.getDatatype()) == DataType.BF16) {
        float[] out = new float[b.remaining() / 2];
        for (int i = 0; i < out.length; i++) {
          out[i] = Float.intBitsToFloat((b.getShort() & 0xFFFF) << 16);
        }
        return out;
      }
      float[] out = new float[b.remaining() / 4];
      b.asFloatBuffer().get(out);
      return out;
    }
    List<Object> d = json(t);
    float[] out = new float[d.size()];
    for (int i = 0; i < out.length; i++) {
      out[i] = ((Number) d.get(i)).floatValue();
    }
    return out;
  }

  public double[] getOutputAsDouble(String name) throws InferenceException {
    IOTensor t = require(name, DataType.FP64);
    ByteBuffer b = binary(name);
    if (b != null) {
      double[] out = new double[b.remaining() / 8];
      b.asDoubleBuffer().get(out);
      return out;
    }
    List<Object> d = json(t);
    double[] out = new double[d.size()];
    for (int i = 0; i < out.length; i++) {
      out[i] = ((Number) d.get(i)).doubleValue();
    }
    return out;
  }

  /** BYTES outputs (also the "value:index[:label]" strings of the classification extension). */
  public String[] getOutputAsString(String name) throws InferenceException {
    IOTensor t = require(name, DataType.BYTES);
    ByteBuffer b = binary(name);
    if (b != null) {
      java.util.ArrayList<String> out = new java.util.ArrayList<>();
      while (b.remaining() >= 4) {
        int n = b.getInt();
        if (n < 0 || n > b.remaining()) {
          throw new InferenceException("malformed BYTES tensor in output '" + name + "'");
        }
        byte[] s = new byte[n];
        b.get(s);
        out.add(new String(s, StandardCharsets.UTF_8));
      }
      return out.toArray(new String[0]);
    }
    List<Object> d = json(t);
    String[] out = new String[d.size()];
    for (int i = 0; i < out.length; i++) {
      out[i] = String.valueOf(d.get(i));
    }
    return out;
  }
}
