// LocalOptimizer.scala — the training loop of tdm/src/main/scala/com/mass/tdm/optim/LocalOptimizer.scala:58-187 with one worker
// per GPU inside ONE JVM (the reference's shape: N worker threads, one model clone each): convertBatch -> negatives sampled on
// the device, trainBatch -> dm_train_forward_backward_dev, syncGradients -> dm_allreduce_grads (RCCL over xGMI), then the same
// dense Adam step on every replica with grad_scale 1 / N.
package com.mass.hip

class LocalOptimizer(engines: Array[HipEngine], layerNegCounts: Array[Int], startSampleLevel: Int, withProb: Boolean,
                     tolerance: Int, useMask: Boolean, learningRate: Double, seqLen: Int) {
  private val n = engines.length
  private val comms = new Array[Long](n)
  engines.foreach(e => Native.trainInit(e.handle, learningRate, 0.0, 0.9, 0.999, 1e-8))        // Adam defaults, Adam.scala:10-16
  if (n > 1) {
    Native.commCreateAll(n, engines.map(_.device), comms)                                       // ncclCommInitAll
    engines.zip(comms).foreach { case (e, c) => Native.commAttach(e.handle, c) }
  }
  private val handles = engines.map(_.handle)

  /** One iteration: worker i trains on its slice (sequences(i): [T_i * seqLen] item ids, targets(i): [T_i]); returns the mean loss. */
  def iteration(sequences: Array[Array[Int]], targets: Array[Array[Int]], seed: Long,
                dSeq: Array[Long], dTgt: Array[Long], dCodes: Array[Long], dSeqs: Array[Long], dMask: Array[Long], dLabels: Array[Long],
                capRows: Long): Double = {
    var lossSum = 0.0
    for (i <- 0 until n) {
      val h = handles(i)
      val t = targets(i).length
      Native.memcpyH2dI32(h, dSeq(i), sequences(i), 4L * t * seqLen)                             // int arrays go up as they are
      Native.memcpyH2dI32(h, dTgt(i), targets(i), 4L * t)
      val rows = new Array[Long](1)
      Native.tdmSampleTrainBatchDev(h, dSeq(i), dTgt(i), t.toLong, seqLen, layerNegCounts, layerNegCounts.length, startSampleLevel,
        if (withProb) 1 else 0, tolerance, if (useMask) 1 else 0, seed + i, dCodes(i), dSeqs(i), dMask(i), dLabels(i), capRows, rows)
      val loss = new Array[Float](1)
      Native.trainForwardBackwardDev(h, dCodes(i), dSeqs(i), dMask(i), dLabels(i), rows(0), seqLen, loss)
      lossSum += loss(0)
    }
    if (n > 1) Native.allreduceGrads(handles, n)                                                 // syncGradients, :164-187
    handles.foreach(h => Native.adamStep(h, 1.0f / n))                                           // ... / realParallelism + Adam.optimize
    lossSum / n
  }
}
