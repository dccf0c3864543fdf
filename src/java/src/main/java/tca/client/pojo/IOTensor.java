package tca.client.pojo;

import java.util.List;

/** One tensor entry of a request or response header: name, datatype, shape, parameters, optional JSON data. */
public class IOTensor {
  private String name;
  private String datatype;
  private long[] shape;
  private Parameters parameters = new Parameters();
  private List<Object> data;  // present when the tensor travels inside the JSON header

  public String getName() {
    return name;
  }

  public void setName(String name) {
    this.name = name;
  }

  public String getDatatype() {
    return datatype;
  }

  public void setDatatype(String datatype) {
    this.datatype = datatype;
  }

  public long[] getShape() {
    return shape;
  }

  public void setShape(long[] shape) {
    this.shape = shape;
  }

  public Parameters getParameters() {
    return parameters;
  }

  public void setParameters(Parameters parameters) {
    this.parameters = parameters == null ? new Parameters() : parameters;
  }

  public List<Object> getData() {
    return data;
  }

  public void setData(List<Object> data) {
    this.data = data;
  }

  /** Number of elements described by the shape. */
  public long elementCount() {
    long n = 1;
    for (long d : shape) {
      n *= d;
    }
    return n;
  }
}
