package tca.client;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.charset.StandardCharsets;

/** Little-endian packing of tensor data for the binary tensor extension of the HTTP protocol. */
public final class BinaryProtocol {
  private BinaryProtocol() {}

  private static ByteBuffer alloc(int bytes) {
    return ByteBuffer.allocate(bytes).order(ByteOrder.LITTLE_ENDIAN);
  }

  public static byte[] toBytes(boolean[] v) {
    byte[] out = new byte[v.length];
    for (int i = 0; i < v.length; i++) {
      out[i] = (byte) (v[i] ? 1 : 0);
    }
    return out;
  }

  public static byte[] toBytes(byte[] v) {
    return v.clone();
  }

  public static byte[] toBytes(short[] v) {
    ByteBuffer b = alloc(v.length * 2);
    b.asShortBuffer().put(v);
    return b.array();
  }

  public static byte[] toBytes(int[] v) {
    ByteBuffer b = alloc(v.length * 4);
    b.asIntBuffer().put(v);
    return b.array();
  }

  public static byte[] toBytes(long[] v) {
    ByteBuffer b = alloc(v.length * 8);
    b.asLongBuffer().put(v);
    return b.array();
  }

  public static byte[] toBytes(float[] v) {
    ByteBuffer b = alloc(v.length * 4);
    b.asFloatBuffer().put(v);
    return b.array();
  }

  public static byte[] toBytes(double[] v) {
    ByteBuffer b = alloc(v.length * 8);
    b.asDoubleBuffer().put(v);
    return b.array();
  }

  /** BYTES tensor: every element is a 4-byte little-endian length followed by the UTF-8 bytes. */
  public static byte[] toBytes(String[] v) {
    byte[][] encoded = new byte[v.length][];
    int total = 0;
    for (int i = 0; i < v.length; i++) {
      encoded[i] = v[i].getBytes(StandardCharsets.UTF_8);
      total += 4 + encoded[i].length;
    }
    ByteBuffer b = alloc(total);
    for (byte[] e : encoded) {
      b.putInt(e.length);
      b.put(e);
    }
    return b.array();
  }

  /** fp32 -> bf16 by truncation (the wire codec's rule), two bytes per element. */
  public static byte[] toBf16Bytes(float[] v) {
    ByteBuffer b = alloc(v.length * 2);
    for (float f : v) {
      b.putShort((short) (Float.floatToRawIntBits(f) >>> 16));
    }
    return b.array();
  }

  public static ByteBuffer view(byte[] data, int offset, int length) {
    return ByteBuffer.wrap(data, offset, length).order(ByteOrder.LITTLE_ENDIAN);
  }
}
