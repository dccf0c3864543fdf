// JavaScript client on stubs built at run time by @grpc/proto-loader from the .proto files of this
// repository: health checks, model metadata and one inference on the "simple" add/sub model with the
// tensors in the typed `contents.int_contents` field.
//
//   npm install && node client.js [host:port]
'use strict';

const path = require('path');

const PROTO_DIR = path.join(__dirname, '..', '..', '..', 'triton_client_amd', 'grpc', 'protos');
const ELEMENTS = 16;

function loadService() {
  const grpc = require('@grpc/grpc-js');
  const protoLoader = require('@grpc/proto-loader');
  const definition = protoLoader.loadSync(path.join(PROTO_DIR, 'grpc_service.proto'), {
    includeDirs: [PROTO_DIR],
    keepCase: true,
    longs: Number,
    enums: String,
    defaults: true,
    oneofs: true,
  });
  return { grpc, inference: grpc.loadPackageDefinition(definition).inference };
}

function call(client, method, request) {
  return new Promise((resolve, reject) => {
    client[method](request, (err, response) => (err ? reject(err) : resolve(response)));
  });
}

function buildRequest(in0, in1) {
  const tensor = (name, values) => ({
    name,
    datatype: 'INT32',
    shape: [1, ELEMENTS],
    contents: { int_contents: values },
  });
  return {
    model_name: 'simple',
    model_version: '',
    inputs: [tensor('INPUT0', in0), tensor('INPUT1', in1)],
    outputs: [{ name: 'OUTPUT0' }, { name: 'OUTPUT1' }],
  };
}

// Responses carry raw little-endian bytes, one buffer per output in order.
function decodeInt32(buffer) {
  const out = [];
  for (let offset = 0; offset + 4 <= buffer.length; offset += 4) {
    out.push(buffer.readInt32LE(offset));
  }
  return out;
}

async function main() {
  const url = process.argv[2] || 'localhost:8001';
  const { grpc, inference } = loadService();
  const client = new inference.GRPCInferenceService(url, grpc.credentials.createInsecure());

  const live = await call(client, 'ServerLive', {});
  const ready = await call(client, 'ServerReady', {});
  if (!live.live || !ready.ready) {
    throw new Error(`server at ${url} is not live/ready`);
  }
  const metadata = await call(client, 'ModelMetadata', { name: 'simple' });
  console.log(`model ${metadata.name}: inputs ${metadata.inputs.map((t) => t.name).join(', ')}`);

  const in0 = Array.from({ length: ELEMENTS }, (_, i) => i);
  const in1 = Array.from({ length: ELEMENTS }, () => 1);
  const response = await call(client, 'ModelInfer', buildRequest(in0, in1));
  if (response.raw_output_contents.length !== 2) {
    throw new Error(`expected 2 raw outputs, got ${response.raw_output_contents.length}`);
  }
  const sum = decodeInt32(response.raw_output_contents[0]);
  const diff = decodeInt32(response.raw_output_contents[1]);
  for (let i = 0; i < ELEMENTS; i += 1) {
    console.log(`${in0[i]} + ${in1[i]} = ${sum[i]}    ${in0[i]} - ${in1[i]} = ${diff[i]}`);
    if (sum[i] !== in0[i] + in1[i] || diff[i] !== in0[i] - in1[i]) {
      throw new Error(`incorrect result at ${i}`);
    }
  }
  client.close();
  console.log('PASS');
}

if (require.main === module) {
  main().catch((err) => {
    console.error(`error: ${err.message}`);
    process.exit(1);
  });
}

module.exports = { buildRequest, decodeInt32 };
