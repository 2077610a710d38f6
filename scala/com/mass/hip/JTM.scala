// JTM.scala — JTM.optimize(): Map[Int, Int] (jtm/src/main/scala/com/mass/jtm/optim/JTM.scala:22-73) as ONE library call: per gap
// step the child weights of every item (TreeLearning.aggregateWeights, jtm/.../optim/TreeLearning.scala:152-174) and the exact greedy
// re-balance of every parent node (:217-265) run on the device, projection and weights stay in HBM between the steps.
// The reference's `numThreads` workers (JTM.scala:33-68: items of a node split while a level has fewer nodes than workers, contiguous
// node ranges afterwards) are GPUs here: pass the engines of this JVM's GPUs as `workers` and the run is sharded over them inside the
// library (RCCL all-gathers over xGMI), with the single-GPU projection bit for bit.
package com.mass.hip

class JTM(engine: HipEngine, itemIds: Array[Int], itemCodes: Array[Int], maxLevel: Int,
          rowOff: Array[Long], rowItemIds: Array[Int], gap: Int, seqLen: Int, hierarchical: Boolean, minLevel: Int,
          useMask: Boolean, workers: Array[HipEngine] = Array.empty) {

  private val engines: Array[HipEngine] = if (workers.isEmpty) Array(engine) else workers
  private val comms = new Array[Long](engines.length)
  if (engines.length > 1) {
    Native.commCreateAll(engines.length, engines.map(_.device), comms)                         // ncclCommInitAll: rank i on engines(i)
    engines.zip(comms).foreach { case (e, c) => Native.commAttach(e.handle, c) }
  }

  def optimize(): Map[Int, Int] = {
    val n = itemIds.length
    val node = new Array[Int](n)
    // every worker's device gets the bookkeeping of all items and the training rows of the item range it scores (JTM.scala:47-52);
    // the loop over the gap steps is one call
    engines.zipWithIndex.foreach { case (e, r) =>
      val lo = new Array[Long](1); val hi = new Array[Long](1)
      Native.jtmShardRange(n.toLong, r, engines.length, lo, hi)
      Native.jtmCacheRowsRange(e.handle, rowOff, rowItemIds, n.toLong, seqLen, lo(0), hi(0))
    }
    try {
      Native.jtmOptimizeAll(engines.map(_.handle), engines.length, itemCodes, n.toLong, maxLevel, gap, if (hierarchical) 1 else 0, minLevel,
        if (useMask) 1 else 0, node, null)
    } finally engines.foreach(e => Native.jtmCacheRows(e.handle, null, null, 0L, seqLen))
    itemIds.zip(node).toMap
  }

  /** Release the communicators of a multi-GPU run (the engines stay the caller's). */
  def close(): Unit = if (engines.length > 1) {
    engines.foreach(e => Native.commAttach(e.handle, 0L))
    comms.foreach(c => if (c != 0L) Native.commDestroy(c))
  }
}
