package tca.client.pojo;

/** Tensor element types of the v2 protocol with their wire size (0 = variable length). */
public enum DataType {
  BOOL(1), UINT8(1), UINT16(2), UINT32(4), UINT64(8), INT8(1), INT16(2), INT32(4), INT64(8), FP16(2), FP32(4),
  FP64(8), BF16(2), BYTES(0);

  private final int byteSize;

  DataType(int byteSize) {
    this.byteSize = byteSize;
  }

  /** Bytes per element on the wire; 0 for BYTES (each element is length-prefixed). */
  public int byteSize() {
    return byteSize;
  }

  public static DataType fromString(String name) {
    return valueOf(name);
  }
}
