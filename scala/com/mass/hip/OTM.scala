// OTM.scala — OTM.recommend (otm/src/main/scala/com/mass/otm/model/OTM.scala:14-22) over dm_otm_beam_search; with f64 weights
// loaded the search runs in the reference's own arithmetic (DIN[Double]).
package com.mass.hip

class OTM(engine: HipEngine, itemIdMapping: Map[Int, Int], useMask: Boolean) extends Serializable {
  val idItemMapping: Map[Int, Int] = itemIdMapping.map(_.swap)
  val leafLevel: Int = math.ceil(math.log(itemIdMapping.size) / math.log(2)).toInt        // upperLog2, otm/package.scala:16
  private val paddingIdx = -1

  def recommend(sequence: Seq[Int], topk: Int, beamSize: Int): Seq[(Int, Double)] = {
    val sequenceIds = sequence.map(itemIdMapping.getOrElse(_, paddingIdx)).toArray
    val ids = new Array[Int](2 * beamSize); val sc = new Array[Double](2 * beamSize); val n = new Array[Int](1)
    Native.otmBeamSearchF64(engine.handle, sequenceIds, 1L, sequenceIds.length, beamSize, leafLevel, ids, sc, n)
    (0 until n(0)).map(i => (ids(i), sc(i)))
      .filter(c => idItemMapping.contains(c._1))
      .sortBy(_._2)(Ordering[Double].reverse)                 // stable, like the reference's sortBy
      .take(topk)
      .map(c => (idItemMapping(c._1), 1.0 / (1 + math.exp(-c._2))))
  }
}

object OTM {
  def apply(engine: HipEngine, itemIdMapping: Map[Int, Int], modelName: String): OTM =
    new OTM(engine, itemIdMapping, modelName.toLowerCase == "din")
}
