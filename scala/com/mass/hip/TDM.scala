// TDM.scala — the reference's facade signatures (tdm/src/main/scala/com/mass/tdm/model/TDM.scala:17-22,
// Recommender.recommendItems tdm/.../model/Recommender.scala:18-37) over the device beam search.
package com.mass.hip

class TDM(engine: HipEngine, useMask: Boolean) extends Serializable {

  /** TDM.predict(sequence, target): Double (TDM.scala:10-15): idToCode over sequence ++ target, one forward, sigmoid in double. */
  def predict(sequence: Array[Int], target: Int): Double = {
    val all = sequence :+ target
    val codes = new Array[Int](all.length); val maskPos = new Array[Int](all.length); val nMask = new Array[Int](1)
    Native.tdmIdToCode(engine.handle, all, all.length, codes, maskPos, nMask)
    val pad = if (useMask) maskPos.take(nMask(0)).filter(_ < sequence.length) else Array.emptyIntArray
    val logit = new Array[Float](1)
    Native.dinForwardF32(engine.handle, Array(codes.last), codes.init, pad, pad.length.toLong, 1L, sequence.length, logit)
    TDM.sigmoid(logit(0))
  }

  /** TDM.recommend(sequence, topk, candidateNum): Array[(Int, Double)] */
  def recommend(sequence: Array[Int], topk: Int, candidateNum: Int): Array[(Int, Double)] = {
    val ids = new Array[Int](topk); val sc = new Array[Float](topk); val n = new Array[Int](1)
    Native.tdmBeamSearch(engine.handle, sequence, 1L, sequence.length, candidateNum, topk, if (useMask) 1 else 0, 0,
      null, null, ids, sc, n)
    Array.tabulate(n(0))(i => (ids(i), TDM.sigmoid(sc(i))))          // sigmoid stays in double, TDM.scala:56-58
  }

  /** Recommender.recommendItems: consumed items dropped, beam widened to max((consumed + topk) / 2, candidateNum). */
  def recommendItems(sequence: Array[Int], topk: Int, candidateNum: Int, consumedItems: Option[Array[Int]]): Array[Int] = {
    val ids = new Array[Int](topk); val sc = new Array[Float](topk); val n = new Array[Int](1)
    val consumed = consumedItems.getOrElse(Array.emptyIntArray)
    Native.tdmBeamSearch(engine.handle, sequence, 1L, sequence.length, candidateNum, topk, if (useMask) 1 else 0,
      if (consumedItems.isDefined) 1 else 0, Array(0L, consumed.length.toLong), consumed, ids, sc, n)
    ids.take(n(0))
  }

  /** One device call for a whole eval batch (Evaluator.evaluate, tdm/.../evaluation/Evaluator.scala:32-71, loops over users). */
  def recommendBatch(sequences: Array[Int], seqLen: Int, topk: Int, candidateNum: Int,
                     consumedOff: Array[Long], consumedIds: Array[Int]): (Array[Int], Array[Float], Array[Int]) = {
    val users = sequences.length / seqLen
    val ids = new Array[Int](users * topk); val sc = new Array[Float](users * topk); val n = new Array[Int](users)
    Native.tdmBeamSearch(engine.handle, sequences, users.toLong, seqLen, candidateNum, topk, if (useMask) 1 else 0,
      if (consumedOff != null) 1 else 0, consumedOff, consumedIds, ids, sc, n)
    (ids, sc, n)
  }
}

object TDM {
  def apply(engine: HipEngine, modelName: String): TDM = new TDM(engine, modelName.toLowerCase == "din")

  /** TDM.saveModel / loadModel (TDM.scala:32-48): one flat checkpoint (weights + index) instead of a Java-serialised module graph. */
  def saveModel(modelPath: String, engine: HipEngine): Unit = Native.saveModel(engine.handle, modelPath)
  def loadModel(engine: HipEngine, modelPath: String, modelName: String): TDM = {
    val name = modelName.toLowerCase
    require(name == "din", "the device scorer is DIN (DeepFM is outside the hot path)")
    Native.loadModel(engine.handle, modelPath)
    new TDM(engine, useMask = true)
  }

  /** TDM.loadTree(treePbPath) (TDM.scala:50-52 -> TDMOp.initTree): the reference's own tree file, parsed by the library. */
  def loadTree(engine: HipEngine, treePbPath: String): Unit = Native.loadTreeFile(engine.handle, treePbPath)
  @inline def sigmoid(logit: Float): Double = 1.0 / (1 + java.lang.Math.exp(-logit))
}
