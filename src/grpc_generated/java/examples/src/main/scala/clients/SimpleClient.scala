package clients

import java.nio.{ByteBuffer, ByteOrder}

import com.google.protobuf.ByteString
import inference.GRPCInferenceServiceGrpc
import inference.GrpcService._
import io.grpc.ManagedChannelBuilder

/** Scala client on the generated Java stubs: the same exchange as SimpleJavaClient, written functionally.
  * Usage: `mvn exec:java -Dexec.mainClass=clients.SimpleClient -Dexec.args="localhost 8001"`.
  */
object SimpleClient {
  private val Elements = 16

  private def pack(values: Seq[Int]): ByteString = {
    val buffer = ByteBuffer.allocate(values.length * 4).order(ByteOrder.LITTLE_ENDIAN)
    values.foreach(buffer.putInt)
    ByteString.copyFrom(buffer.array())
  }

  private def unpack(raw: ByteString): Vector[Int] = {
    val view = raw.asReadOnlyByteBuffer().order(ByteOrder.LITTLE_ENDIAN).asIntBuffer()
    Vector.fill(view.remaining())(view.get())
  }

  private def tensor(name: String): ModelInferRequest.InferInputTensor =
    ModelInferRequest.InferInputTensor.newBuilder().setName(name).setDatatype("INT32").addShape(1).addShape(Elements).build()

  def main(args: Array[String]): Unit = {
    val host = args.headOption.getOrElse("localhost")
    val port = args.lift(1).map(_.toInt).getOrElse(8001)
    val channel = ManagedChannelBuilder.forAddress(host, port).usePlaintext().build()
    try {
      val stub = GRPCInferenceServiceGrpc.newBlockingStub(channel)
      require(stub.serverLive(ServerLiveRequest.getDefaultInstance).getLive, "server is not live")
      require(stub.serverReady(ServerReadyRequest.getDefaultInstance).getReady, "server is not ready")
      val metadata = stub.modelMetadata(ModelMetadataRequest.newBuilder().setName("simple").build())
      println(s"model ${metadata.getName}: ${metadata.getInputsCount} inputs, ${metadata.getOutputsCount} outputs")

      val in0 = (0 until Elements).toVector
      val in1 = Vector.fill(Elements)(1)
      val request = ModelInferRequest.newBuilder()
        .setModelName("simple")
        .addInputs(tensor("INPUT0"))
        .addInputs(tensor("INPUT1"))
        .addOutputs(ModelInferRequest.InferRequestedOutputTensor.newBuilder().setName("OUTPUT0"))
        .addOutputs(ModelInferRequest.InferRequestedOutputTensor.newBuilder().setName("OUTPUT1"))
        .addRawInputContents(pack(in0))
        .addRawInputContents(pack(in1))
        .build()
      val response = stub.modelInfer(request)
      val sum = unpack(response.getRawOutputContents(0))
      val diff = unpack(response.getRawOutputContents(1))
      in0.indices.foreach { i =>
        println(s"${in0(i)} + ${in1(i)} = ${sum(i)}    ${in0(i)} - ${in1(i)} = ${diff(i)}")
        require(sum(i) == in0(i) + in1(i) && diff(i) == in0(i) - in1(i), s"incorrect result at $i")
      }
      println("PASS")
    } finally {
      channel.shutdownNow()
    }
  }
}
