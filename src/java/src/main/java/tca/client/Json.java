package tca.client;

import java.util.ArrayList;
import java.util.LinkedHashMap;
import java.util.List;
import java.util.Map;

/**
 * Minimal JSON codec for the protocol headers: objects become {@code Map<String,Object>}, arrays
 * {@code List<Object>}, numbers {@code Long} or {@code Double}.  Accepts NaN / Infinity literals, which
 * servers may emit for floating point tensors.
 */
public final class Json {
  private Json() {}

  public static Object parse(String text) throws InferenceException {
    Parser p = new Parser(text);
    p.skipWs();
    Object v = p.value();
    p.skipWs();
    if (p.pos != text.length()) {
      throw p.fail("trailing characters");
    }
    return v;
  }

  public static String write(Object value) {
    StringBuilder sb = new StringBuilder();
    write(sb, value);
    return sb.toString();
  }

  @SuppressWarnings("unchecked")
  private static void write(StringBuilder sb, Object v) {
    if (v == null) {
      sb.append("null");
    } else if (v instanceof String) {
      quote(sb, (String) v);
    } else if (v instanceof Boolean || v instanceof Long || v instanceof Integer || v instanceof Short || v instanceof Byte) {
      sb.append(v);
    } else if (v instanceof Number) {
      double d = ((Number) v).doubleValue();
      if (Double.isNaN(d)) {
        sb.append("NaN");
      } else if (Double.isInfinite(d)) {
        sb.append(d < 0 ? "-Infinity" : "Infinity");
      } else {
        sb.append(v);
      }
    } else if (v instanceof Map) {
      sb.append('{');
      boolean first = true;
      for (Map.Entry<String, Object> e : ((Map<String, Object>) v).entrySet()) {
        if (!first) {
          sb.append(',');
        }
        first = false;
        quote(sb, e.getKey());
        sb.append(':');
        write(sb, e.getValue());
      }
      sb.append('}');
    } else if (v instanceof Iterable) {
      sb.append('[');
      boolean first = true;
      for (Object item : (Iterable<Object>) v) {
        if (!first) {
          sb.append(',');
        }
        first = false;
        write(sb, item);
      }
      sb.append(']');
    } else if (v instanceof long[]) {
      sb.append('[');
      long[] a = (long[]) v;
      for (int i = 0; i < a.length; i++) {
        if (i > 0) {
          sb.append(',');
        }
        sb.append(a[i]);
      }
      sb.append(']');
    } else {
      throw new IllegalArgumentException("cannot serialise " + v.getClass().getName());
    }
  }

  private static void quote(StringBuilder sb, String s) {
    sb.append('"');
    for (int i = 0; i < s.length(); i++) {
      char c = s.charAt(i);
      switch (c) {
        case '"': sb.append("\\\""); break;
        case '\\': sb.append("\\\\"); break;
        case '\n': sb.append("\\n"); break;
        case '\r': sb.append("\\r"); break;
        case '\t': sb.append("\\t"); break;
        case '\b': sb.append("\\b"); break;
        case '\f': sb.append("\\f"); break;
        default:
          if (c < 0x20) {
            sb.append(String.format("\\u%04x", (int) c));
          } else {
            sb.append(c);
          }
      }
    }
    sb.append('"');
  }

  private static final class Parser {
    final String s;
    int pos = 0;

    Parser(String s) {
      this.s = s;
    }

    InferenceException fail(String what) {
      return new InferenceException("malformed JSON at offset " + pos + ": " + what);
    }

    void skipWs() {
      while (pos < s.length() && Character.isWhitespace(s.charAt(pos))) {
        pos++;
      }
    }

    boolean match(String lit) {
      if (s.startsWith(lit, pos)) {
        pos += lit.length();
        return true;
      }
      return false;
    }

    Object value() throws InferenceException {
      if (pos >= s.length()) {
        throw fail("unexpected end");
      }
      char c = s.charAt(pos);
      if (c == '{') {
        return object();
      }
      if (c == '[') {
        return array();
      }
      if (c == '"') {
        return string();
      }
      if (match("true")) {
        return Boolean.TRUE;
      }
      if (match("false")) {
        return Boolean.FALSE;
      }
      if (match("null")) {
        return null;
      }
      if (match("NaN")) {
        return Double.NaN;
      }
      if (match("Infinity")) {
        return Double.POSITIVE_INFINITY;
      }
      if (match("-Infinity")) {
        return Double.NEGATIVE_INFINITY;
      }
      return number();
    }

    Object number() throws InferenceException {
      int start = pos;
      boolean integral = true;
      while (pos < s.length()) {
        char c = s.charAt(pos);
        if (c == '.' || c == 'e' || c == 'E') {
          integral = false;
        } else if (!(Character.isDigit(c) || c == '-' || c == '+')) {
          break;
        }
        pos++;
      }
      if (start == pos) {
        throw fail("unexpected character '" + s.charAt(pos) + "'");
      }
      String text = s.substring(start, pos);
      try {
        if (integral) {
          try {
            return Long.parseLong(text);
          } catch (NumberFormatException tooBig) {
            return Double.parseDouble(text);  // uint64 beyond Long.MAX_VALUE
          }
        }
        return Double.parseDouble(text);
      } catch (NumberFormatException e) {
        throw fail("bad number '" + text + "'");
      }
    }

    String string() throws InferenceException {
      pos++;  // opening quote
      StringBuilder sb = new StringBuilder();
      while (pos < s.length()) {
        char c = s.charAt(pos++);
        if (c == '"') {
          return sb.toString();
        }
        if (c != '\\') {
          sb.append(c);
          continue;
        }
        if (pos >= s.length()) {
          break;
        }
        char e = s.charAt(pos++);
        switch (e) {
          case 'n': sb.append('\n'); break;
          case 'r': sb.append('\r'); break;
          case 't': sb.append('\t'); break;
          case 'b': sb.append('\b'); break;
          case 'f': sb.append('\f'); break;
          case 'u':
            if (pos + 4 > s.length()) {
              throw fail("truncated \\u escape");
            }
            sb.append((char) Integer.parseInt(s.substring(pos, pos + 4), 16));
            pos += 4;
            break;
          default: sb.append(e);
        }
      }
      throw fail("unterminated string");
    }

    List<Object> array() throws InferenceException {
      pos++;
      List<Object> out = new ArrayList<>();
      skipWs();
      if (pos < s.length() && s.charAt(pos) == ']') {
        pos++;
        return out;
      }
      while (true) {
        skipWs();
        out.add(value());
        skipWs();
        if (pos >= s.length()) {
          throw fail("unterminated array");
        }
        char c = s.charAt(pos++);
        if (c == ']') {
          return out;
        }
        if (c != ',') {
          throw fail("expected ',' or ']'");
        }
      }
    }

    Map<String, Object> object() throws InferenceException {
      pos++;
      Map<String, Object> out = new LinkedHashMap<>();
      skipWs();
      if (pos < s.length() && s.charAt(pos) == '}') {
        pos++;
        return out;
      }
      while (true) {
        skipWs();
        if (pos >= s.length() || s.charAt(pos) != '"') {
          throw fail("expected a member name");
        }
        String key = string();
        skipWs();
        if (pos >= s.length() || s.charAt(pos++) != ':') {
          throw fail("expected ':'");
        }
        skipWs();
        out.put(key, value());
        skipWs();
        if (pos >= s.length()) {
          throw fail("unterminated object");
        }
        char c = s.charAt(pos++);
        if (c == '}') {
          return out;
        }
        if (c != ',') {
          throw fail("expected ',' or '}'");
        }
      }
    }
  }
}
