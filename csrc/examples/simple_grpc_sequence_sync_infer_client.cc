// Stateful sequences with synchronous gRPC requests; -d selects the dyna
// models.  (Reference: `src/c++/examples/simple_grpc_sequence_sync_infer_client.cc`.)
#include "example_utils.h"

static int32_t Send(tc::InferenceServerGrpcClient* client, const std::string& model, uint64_t id,
                    const std::string& id_str, int32_t value, bool start, bool end) {
  auto in = NewInput("INPUT", {1, 1}, "INT32");
  in->AppendRaw(reinterpret_cast<const uint8_t*>(&value), sizeof(value));
  tc::InferOptions options(model);
  options.sequence_id_ = id;
  options.sequence_id_str_ = id_str;
  options.sequence_start_ = start;
  options.sequence_end_ = end;
  tc::InferResult* result = nullptr;
  FAIL_IF_ERR(client->Infer(&result, options, {in.get()}), "sequence request");
  const uint8_t* buf = nullptr;
  size_t n = 0;
  FAIL_IF_ERR(result->RawData("OUTPUT", &buf, &n), "OUTPUT");
  const int32_t out = tca::LoadAs<int32_t>(buf);
  delete result;
  return out;
}

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  bool dyna = false;
  for (const auto& a : flags.rest) dyna |= a == "-d";
  auto client = MakeGrpcClient(flags);
  const std::string int_model = dyna ? "simple_dyna_sequence" : "simple_sequence";
  const std::string str_model = dyna ? "simple_string_dyna_sequence" : "simple_sequence";
  const std::vector<int32_t> values = {11, 7, 5, 3, 2, 0, 1};
  const uint64_t id0 = 1000, id1 = 1001;
  const std::string sid = dyna ? "1002" : "a-string-correlation-id";
  std::vector<int32_t> r0, r1, rs;
  for (size_t i = 0; i <= values.size(); ++i) {
    const bool start = i == 0, end = i == values.size();
    r0.push_back(Send(client.get(), int_model, id0, "", start ? 0 : values[i - 1], start, end));
    r1.push_back(Send(client.get(), int_model, id1, "", start ? 100 : -values[i - 1], start, end));
    rs.push_back(Send(client.get(), str_model, 0, sid, start ? 20 : -values[i - 1], start, end));
  }
  for (size_t i = 0; i <= values.size(); ++i) {
    int32_t e0 = i == 0 ? 1 : values[i - 1];
    int32_t e1 = i == 0 ? 101 : -values[i - 1];
    int32_t es = dyna ? (i == 0 ? 20 : rs[i - 1] - values[i - 1]) : (i == 0 ? 21 : -values[i - 1]);
    if (dyna && i == values.size()) {
      e0 += static_cast<int32_t>(id0);
      e1 += static_cast<int32_t>(id1);
      es += 1002;
    }
    std::cout << "[" << i << "] " << r0[i] << " : " << r1[i] << " : " << rs[i] << std::endl;
    if (r0[i] != e0 || r1[i] != e1 || rs[i] != es) FAIL("[ expected ] " << e0 << " : " << e1 << " : " << es);
  }
  std::cout << "PASS : Sequence" << std::endl;
  return 0;
}
