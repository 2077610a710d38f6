// TDM.scala — the reference's facade signatures (tdm/src/main/scala/com/mass/tdm/model/TDM.scala:17-22,
// Recommender.recommendItems tdm/.../model/Recommender.scala:18-37) over the device beam search.
package com.mass.hip

class TDM(engine: HipEngine, useMask: Boolean) extends Serializable {

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
  @inline def sigmoid(logit: Float): Double = 1.0 / (1 + java.lang.Math.exp(-logit))
}
