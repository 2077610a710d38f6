// DeepRetrieval.scala — DeepRetrieval.recommend (deep-retrieval/src/main/scala/com/mass/dr/model/DeepRetrieval.scala:26-46).
package com.mass.hip

class DeepRetrieval(engine: HipEngine, itemIdMapping: Map[Int, Int]) extends Serializable {
  val idItemMapping: Map[Int, Int] = itemIdMapping.map(_.swap)          // MappingOp.scala:16
  private val paddingIdx = -1

  /** LayerModel / RerankModel storage arrays (fp64) + MappingOp.pathItemMapping flattened to a CSR over distinct paths. */
  def load(embed: Int, seqLen: Int, numNode: Int, numLayer: Int, numItem: Long, layerEmb: Array[Double],
           layerW: Array[Array[Double]], layerB: Array[Array[Double]], rerankEmb: Array[Double], rerankW: Array[Double],
           rerankB: Array[Double], softmaxW: Array[Double], softmaxB: Array[Double],
           pathNodes: Array[Int], itemOff: Array[Long], items: Array[Int]): Unit = {
    Native.drLoadModelF64(engine.handle, embed, seqLen, numNode, numLayer, numItem, layerEmb, layerW, layerB, rerankEmb, rerankW,
      rerankB, softmaxW, softmaxB)
    Native.drLoadPathItems(engine.handle, pathNodes, (itemOff.length - 1).toLong, itemOff, items)
  }

  def recommend(sequence: Seq[Int], topk: Int, beamSize: Int): Seq[(Int, Double)] = {
    val sequenceIds = sequence.map(itemIdMapping.getOrElse(_, paddingIdx)).toArray
    val ids = new Array[Int](topk); val sc = new Array[Double](topk); val n = new Array[Int](1)
    Native.drRecommend(engine.handle, sequenceIds, 1L, beamSize, topk, ids, sc, n)
    (0 until n(0)).map(i => (idItemMapping(ids(i)), 1.0 / (1 + math.exp(-sc(i)))))
  }
}
