// HipEngine.scala — one dm_handle_t (device + stream) behind a Scala object, plus the `.conf` reader the tasks use.
// Hand-written over the generated Native object; source only in the build image (no JVM there), kept honest by
// tests/test_jni_shim.py: every Native.<method> used below must exist in Native.scala with that arity.
package com.mass.hip

/** Property.readConf / getOrStop (scalann/src/main/scala/com/mass/scalann/utils/Property.scala:12-71) are unchanged in the
  * reference and keep reading configs/*.conf; this engine only replaces what the tasks do with the values. */
final class HipEngine private (val device: Int, adopted: Long) extends AutoCloseable {
  def this(device: Int = 0) = this(device, 0L)
  val handle: Long = if (adopted != 0L) adopted else { val out = new Array[Long](1); Native.create(device, out); out(0) }
  private var embed = 0
  private var maxLevel_ = 0

  /** Module.cloneModule() as the reference's worker threads use it (tdm/.../optim/LocalOptimizer.scala:28-44, pinned by
    * otm/src/test/scala/CloneModelSpec.scala:20-36): an engine that READS this engine's tree and weights — the same device memory,
    * dm_clone copies nothing — through its own stream and request buffers, one per serving thread.  Loading and training stay with
    * the owner; close the clones first. */
  def cloneEngine(): HipEngine = {
    val out = new Array[Long](1)
    Native.clone(handle, out)
    val c = new HipEngine(device, out(0))
    c.embed = embed; c.maxLevel_ = maxLevel_
    c
  }

  def maxLevel: Int = maxLevel_
  def embedSize: Int = embed

  /** DistTree.loadData's result (tdm/.../tree/DistTree.scala:40-87): codeNodeMap as parallel arrays + the id -> code pairs. */
  def loadTree(codes: Array[Int], nodeIds: Array[Int], isLeaf: Array[Byte], maxLevel: Int,
               leafItemIds: Array[Int], leafCodes: Array[Int]): Unit = {
    Native.loadTreeTdm(handle, codes, nodeIds, isLeaf, codes.length.toLong, maxLevel)
    Native.loadIdMaps(handle, leafItemIds, leafCodes, leafItemIds.length.toLong)
    maxLevel_ = maxLevel
  }

  /** Node.probality per node, for model.sample_with_probability (NegativeSampler.levelProbs, NegativeSampler.scala:59-66). */
  def setNodeProbs(codes: Array[Int], probs: Array[Float]): Unit =
    Native.tdmSetNodeProbs(handle, codes, probs, codes.length.toLong)

  /** The flat array behind Module.parameters() (Graph.parameters order, scalann/.../nn/graphnn/Graph.scala:37-48). */
  def loadWeights(compact: Array[Float], embedSize: Int, numIndex: Long): Unit = {
    Native.loadWeightsDinF32(handle, embedSize, numIndex, compact, compact.length.toLong); embed = embedSize
  }
  def loadWeights(compact: Array[Double], embedSize: Int, numIndex: Long): Unit = {
    Native.loadWeightsDinF64(handle, embedSize, numIndex, compact, compact.length.toLong); embed = embedSize
  }

  /** conf key `model.scorer`: 0 f32, 1 split_f16, 2 auto (default), 3 f64 (dismember_hip.h: dm_set_scorer_mode). */
  def setScorerMode(mode: Int): Unit = Native.setScorerMode(handle, mode)

  /** Module.forward(Table(items, seqs, masks)) for f32 models (T/model/Recommender.scala:93-94). */
  def forward(codes: Array[Int], seqs: Array[Int], padFlatIdx: Array[Int], seqLen: Int): Array[Float] = {
    val logits = new Array[Float](codes.length)
    Native.dinForwardF32(handle, codes, seqs, padFlatIdx, padFlatIdx.length.toLong, codes.length.toLong, seqLen, logits)
    logits
  }

  override def close(): Unit = Native.destroy(handle)
}
