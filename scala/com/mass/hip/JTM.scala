// JTM.scala — JTM.optimize(): Map[Int, Int] (jtm/src/main/scala/com/mass/jtm/optim/JTM.scala:22-73): per gap step the child
// weights of every item on the device (TreeLearning.aggregateWeights, jtm/.../optim/TreeLearning.scala:152-174) and the exact
// greedy re-balance of every parent node (:217-265) in one call each.
package com.mass.hip

class JTM(engine: HipEngine, itemIds: Array[Int], itemCodes: Array[Int], maxLevel: Int,
          rowOff: Array[Long], rowItemIds: Array[Int], gap: Int, seqLen: Int, hierarchical: Boolean, minLevel: Int,
          useMask: Boolean) {

  private def ancestorAtLevel(code: Int, level: Int): Int = {              // JTMTree.getAncestorAtLevel (JTMTree.scala:36-43)
    var c = code
    val depth = 31 - Integer.numberOfLeadingZeros(c + 1)
    var d = depth
    while (d > level) { c = (c - 1) >> 1; d -= 1 }
    c
  }

  def optimize(): Map[Int, Int] = {
    val n = itemIds.length
    var node = new Array[Int](n)                                           // every item starts at the root
    // itemSequenceMap goes to the device once; the loop over the gap steps is one call (scoring + greedy re-balance of every parent node
    // on the device, projection and weights stay in HBM between the steps)
    Native.jtmCacheRows(engine.handle, rowOff, rowItemIds, n.toLong, seqLen)
    try {
      Native.jtmOptimizeCached(engine.handle, itemCodes, n.toLong, maxLevel, gap, if (hierarchical) 1 else 0, minLevel,
        if (useMask) 1 else 0, node, null)
    } finally Native.jtmCacheRows(engine.handle, null, null, 0L, seqLen)
    itemIds.zip(node).toMap
  }
}
