package tca.client;

import java.net.URLEncoder;
import java.nio.charset.StandardCharsets;
import java.util.ArrayList;
import java.util.List;
import java.util.Map;
import tca.client.pojo.IOTensor;
import tca.client.pojo.InferenceResponse;
import tca.client.pojo.Parameters;

/** Small helpers shared by the client classes. */
public final class Util {
  private Util() {}

  public static boolean isEmpty(String s) {
    return s == null || s.isEmpty();
  }

  public static String urlEncode(String s) {
    return URLEncoder.encode(s, StandardCharsets.UTF_8).replace("+", "%20");
  }

  public static long elementCount(long[] shape) {
    long n = 1;
    for (long d : shape) {
      n *= d;
    }
    return n;
  }

  public static long[] toLongArray(List<Object> values) {
    long[] out = new long[values.size()];
    for (int i = 0; i < out.length; i++) {
      out[i] = ((Number) values.get(i)).longValue();
    }
    return out;
  }

  @SuppressWarnings("unchecked")
  static InferenceResponse responseFromJson(Object parsed) throws InferenceException {
    if (!(parsed instanceof Map)) {
      throw new InferenceException("response header is not a JSON object");
    }
    Map<String, Object> m = (Map<String, Object>) parsed;
    InferenceResponse r = new InferenceResponse();
    r.setModelName((String) m.get("model_name"));
    r.setModelVersion((String) m.get("model_version"));
    r.setId((String) m.get("id"));
    if (m.get("parameters") instanceof Map) {
      r.setParameters(new Parameters((Map<String, Object>) m.get("parameters")));
    }
    List<IOTensor> outputs = new ArrayList<>();
    if (m.get("outputs") instanceof List) {
      for (Object o : (List<Object>) m.get("outputs")) {
        Map<String, Object> t = (Map<String, Object>) o;
        IOTensor tensor = new IOTensor();
        tensor.setName((String) t.get("name"));
        tensor.setDatatype((String) t.get("datatype"));
        tensor.setShape(toLongArray((List<Object>) t.get("shape")));
        if (t.get("parameters") instanceof Map) {
          tensor.setParameters(new Parameters((Map<String, Object>) t.get("parameters")));
        }
        if (t.get("data") instanceof List) {
          tensor.setData(flatten((List<Object>) t.get("data"), new ArrayList<>()));
        }
        outputs.add(tensor);
      }
    }
    r.setOutputs(outputs);
    return r;
  }

  @SuppressWarnings("unchecked")
  private static List<Object> flatten(List<Object> nested, List<Object> out) {
    for (Object v : nested) {
      if (v instanceof List) {
        flatten((List<Object>) v, out);
      } else {
        out.add(v);
      }
    }
    return out;
  }
}
