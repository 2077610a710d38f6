// OTMLocalOptimizer.scala — the training loop body of otm/src/main/scala/com/mass/otm/optim/LocalOptimizer.scala:55-109 with one worker
// per GPU inside ONE JVM (the reference's shape: numThread model clones): per batch and worker ONE native call (dm_otm_train_batch) builds
// the pseudo targets (OTMTree.optimalPseudoTargets, otm/.../tree/OTMTree.scala:27-46), the beam nodes (beamSearchNodes, :67-91) and every
// level's MiniBatch on the device and runs forward / backward, syncGradients (RCCL over xGMI inside the library) and Adam per level.
package com.mass.hip

import scala.concurrent.{Await, Future}
import scala.concurrent.duration.Duration
import scala.concurrent.ExecutionContext.Implicits.global

class OTMLocalOptimizer(engines: Array[HipEngine], leafLevel: Int, beamSize: Int, seqLen: Int, learningRate: Double, useMask: Boolean,
                        targetMode: String = "pseudo") {
  private val n = engines.length
  private val comms = new Array[Long](n)
  engines.foreach(e => Native.trainInit(e.handle, learningRate, 0.0, 0.9, 0.999, 1e-8))        // Adam defaults, Adam.scala:10-16
  if (n > 1) {
    Native.commCreateAll(n, engines.map(_.device), comms)                                       // ncclCommInitAll
    engines.zip(comms).foreach { case (e, c) => Native.commAttach(e.handle, c) }
  }
  private val startLevel = 31 - Integer.numberOfLeadingZeros(beamSize)                          // lowerLog2, otm/package.scala:15
  val numLevels: Int = leafLevel - startLevel

  /** One iteration: worker i trains on its slice — sequences(i): [U_i * seqLen] node ids (-1 = padding), targetOff(i): [U_i + 1] and
    * targetNodes(i): the CSR of its users' target leaf nodes.  The call is collective across the workers (one gradient exchange per
    * level), so every worker runs on its own thread.  Returns the per-level losses (already averaged over the workers). */
  def iteration(sequences: Array[Array[Int]], targetOff: Array[Array[Long]], targetNodes: Array[Array[Int]]): Array[Double] = {
    val runs = (0 until n).map { i =>
      Future {
        val losses = new Array[Double](numLevels)
        val nl = new Array[Int](1)
        Native.otmTrainBatch(engines(i).handle, sequences(i), (targetOff(i).length - 1).toLong, seqLen, targetOff(i), targetNodes(i),
          beamSize, leafLevel, if (useMask) 1 else 0, if (targetMode == "normal") 1 else 0, losses, nl)
        losses.take(nl(0))
      }
    }
    runs.map(Await.result(_, Duration.Inf)).head
  }

  def close(): Unit = if (n > 1) {
    engines.foreach(e => Native.commAttach(e.handle, 0L))
    comms.foreach(c => if (c != 0L) Native.commDestroy(c))
  }
}
