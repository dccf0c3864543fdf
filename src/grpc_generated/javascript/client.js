// JavaScript client: health checks, model metadata and one inference on the "simple" add/sub model
// with the tensors in the typed `contents.int_contents` field.
//
// Two transports behind the same calls:
//   * stubs built at run time by @grpc/proto-loader from the .proto files of this repository, when
//     @grpc/grpc-js is installed (`npm install`), as the reference's demo does;
//   * otherwise node's own `http2` module with the few messages of the demo coded by hand
//     (protobuf varints and length-delimited fields, gRPC 5-byte message framing) - no packages
//     needed, which is also how the demo is run by this repository's tests.
//
//   node client.js [host:port] [--transport grpc-js|core]
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

// ---- protobuf wire format, as much as the demo's messages need -------------------------------
function varint(value) {
  const bytes = [];
  let v = BigInt.asUintN(64, BigInt(value));
  do {
    let b = Number(v & 0x7fn);
    v >>= 7n;
    if (v > 0n) b |= 0x80;
    bytes.push(b);
  } while (v > 0n);
  return Buffer.from(bytes);
}

const tag = (field, wireType) => varint((field << 3) | wireType);
const lenField = (field, payload) => Buffer.concat([tag(field, 2), varint(payload.length), payload]);
const strField = (field, text) => (text ? lenField(field, Buffer.from(text, 'utf8')) : Buffer.alloc(0));
const packedField = (field, values) => lenField(field, Buffer.concat(values.map((v) => varint(v))));

// Decodes one message into { fieldNumber: [values...] }: varints as Number, length-delimited as Buffer.
function decodeMessage(buffer) {
  const fields = {};
  let pos = 0;
  const readVarint = () => {
    let result = 0n;
    let shift = 0n;
    for (;;) {
      if (pos >= buffer.length) throw new Error('truncated protobuf message');
      const b = buffer[pos++];
      result |= BigInt(b & 0x7f) << shift;
      if (!(b & 0x80)) return result;
      shift += 7n;
    }
  };
  while (pos < buffer.length) {
    const key = Number(readVarint());
    const field = key >> 3;
    let value;
    switch (key & 7) {
      case 0: value = Number(BigInt.asIntN(64, readVarint())); break;
      case 1: value = buffer.slice(pos, pos + 8); pos += 8; break;
      case 2: { const n = Number(readVarint()); value = buffer.slice(pos, pos + n); pos += n; break; }
      case 5: value = buffer.slice(pos, pos + 4); pos += 4; break;
      default: throw new Error(`unsupported wire type ${key & 7}`);
    }
    (fields[field] = fields[field] || []).push(value);
  }
  return fields;
}

function encodeInferRequest(request) {
  const parts = [strField(1, request.model_name), strField(2, request.model_version), strField(3, request.id)];
  for (const t of request.inputs) {
    const contents = packedField(2, t.contents.int_contents);  // InferTensorContents.int_contents
    parts.push(lenField(5, Buffer.concat([strField(1, t.name), strField(2, t.datatype), packedField(3, t.shape),
      lenField(5, contents)])));
  }
  for (const o of request.outputs || []) parts.push(lenField(6, strField(1, o.name)));
  return Buffer.concat(parts);
}

function decodeTensorNames(buffers) {
  return (buffers || []).map((b) => ({ name: (decodeMessage(b)[1] || [Buffer.alloc(0)])[0].toString('utf8') }));
}

const CODECS = {
  ServerLive: { encode: () => Buffer.alloc(0), decode: (f) => ({ live: Boolean((f[1] || [0])[0]) }) },
  ServerReady: { encode: () => Buffer.alloc(0), decode: (f) => ({ ready: Boolean((f[1] || [0])[0]) }) },
  ModelMetadata: {
    encode: (r) => Buffer.concat([strField(1, r.name), strField(2, r.version)]),
    decode: (f) => ({
      name: (f[1] || [Buffer.alloc(0)])[0].toString('utf8'),
      platform: (f[3] || [Buffer.alloc(0)])[0].toString('utf8'),
      inputs: decodeTensorNames(f[4]),
      outputs: decodeTensorNames(f[5]),
    }),
  },
  ModelInfer: {
    encode: encodeInferRequest,
    decode: (f) => ({
      model_name: (f[1] || [Buffer.alloc(0)])[0].toString('utf8'),
      outputs: decodeTensorNames(f[5]),
      raw_output_contents: f[6] || [],
    }),
  },
};

// gRPC over node's http2: one stream per call, 5-byte prefix (compressed flag + big-endian length).
class CoreClient {
  constructor(url) {
    this.session = require('http2').connect(`http://${url}`);
    this.session.on('error', () => {});
  }

  call(method, request) {
    const codec = CODECS[method];
    return new Promise((resolve, reject) => {
      const payload = codec.encode(request);
      const prefix = Buffer.alloc(5);
      prefix.writeUInt32BE(payload.length, 1);
      const stream = this.session.request({
        ':method': 'POST',
        ':path': `/inference.GRPCInferenceService/${method}`,
        'content-type': 'application/grpc',
        te: 'trailers',
      });
      const chunks = [];
      let status = null;
      let message = '';
      const take = (headers) => {
        if (headers['grpc-status'] !== undefined) {
          status = Number(headers['grpc-status']);
          message = decodeURIComponent(headers['grpc-message'] || '');
        }
      };
      stream.on('response', take);
      stream.on('trailers', take);
      stream.on('data', (chunk) => chunks.push(chunk));
      stream.on('error', reject);
      stream.on('end', () => {
        if (status !== 0) {
          reject(new Error(`gRPC status ${status}: ${message}`));
          return;
        }
        const body = Buffer.concat(chunks);
        if (body.length < 5 || body.readUInt32BE(1) !== body.length - 5) {
          reject(new Error('malformed gRPC response'));
          return;
        }
        resolve(codec.decode(decodeMessage(body.slice(5))));
      });
      stream.end(Buffer.concat([prefix, payload]));
    });
  }

  close() {
    this.session.close();
  }
}

function connect(url, transport) {
  if (transport !== 'core') {
    try {
      const { grpc, inference } = loadService();
      const client = new inference.GRPCInferenceService(url, grpc.credentials.createInsecure());
      return { call: (method, request) => call(client, method, request), close: () => client.close(), transport: 'grpc-js' };
    } catch (err) {
      if (transport === 'grpc-js') throw err;
    }
  }
  const core = new CoreClient(url);
  return { call: (method, request) => core.call(method, request), close: () => core.close(), transport: 'core' };
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
  const args = process.argv.slice(2);
  let transport = 'auto';
  const at = args.indexOf('--transport');
  if (at >= 0) {
    transport = args[at + 1];
    args.splice(at, 2);
  }
  const url = args[0] || 'localhost:8001';
  const client = connect(url, transport);

  const live = await client.call('ServerLive', {});
  const ready = await client.call('ServerReady', {});
  if (!live.live || !ready.ready) {
    throw new Error(`server at ${url} is not live/ready`);
  }
  const metadata = await client.call('ModelMetadata', { name: 'simple' });
  console.log(`model ${metadata.name}: inputs ${metadata.inputs.map((t) => t.name).join(', ')} (transport: ${client.transport})`);

  const in0 = Array.from({ length: ELEMENTS }, (_, i) => i);
  const in1 = Array.from({ length: ELEMENTS }, () => 1);
  const response = await client.call('ModelInfer', buildRequest(in0, in1));
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
  // an unknown model comes back as a gRPC status, not as a broken call
  let refused = false;
  try {
    await client.call('ModelMetadata', { name: 'no_such_model' });
  } catch (err) {
    refused = true;
  }
  if (!refused) {
    throw new Error('metadata of an unknown model was not refused');
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

module.exports = { buildRequest, decodeInt32, encodeInferRequest, decodeMessage, varint };
