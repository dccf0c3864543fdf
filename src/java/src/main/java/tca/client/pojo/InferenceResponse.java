package tca.client.pojo;

import java.util.ArrayList;
import java.util.List;

/** JSON header of an inference response. */
public class InferenceResponse {
  private String modelName;
  private String modelVersion;
  private String id;
  private Parameters parameters = new Parameters();
  private List<IOTensor> outputs = new ArrayList<>();

  public String getModelName() {
    return modelName;
  }

  public void setModelName(String modelName) {
    this.modelName = modelName;
  }

  public String getModelVersion() {
    return modelVersion;
  }

  public void setModelVersion(String modelVersion) {
    this.modelVersion = modelVersion;
  }

  public String getId() {
    return id;
  }

  public void setId(String id) {
    this.id = id;
  }

  public Parameters getParameters() {
    return parameters;
  }

  public void setParameters(Parameters parameters) {
    this.parameters = parameters;
  }

  public List<IOTensor> getOutputs() {
    return outputs;
  }

  public void setOutputs(List<IOTensor> outputs) {
    this.outputs = outputs;
  }

  public IOTensor getOutputByName(String name) {
    for (IOTensor t : outputs) {
      if (t.getName().equals(name)) {
        return t;
      }
    }
    return null;
  }
}
