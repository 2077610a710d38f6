"""Mirrors of the reference's L7 facades for the retrieval path.

  TDM  <- tdm/src/main/scala/com/mass/tdm/model/TDM.scala:8-58 (+ Recommender.recommendItems,
          tdm/.../model/Recommender.scala:18-37)
  OTM  <- otm/src/main/scala/com/mass/otm/model/OTM.scala:8-61

Same names, argument meaning and result shapes; the level loop and the scorer run in
libdismember_hip.so.  Sequences may be one user's list (reference signature) or a [U, L]
batch (what the device is built for).
"""
import numpy as np

from .engine import Engine


def sigmoid(logit):
    """TDM.sigmoid: computed in double (TDM.scala:56-58)."""
    return 1.0 / (1.0 + np.exp(-np.asarray(logit, dtype=np.float64)))


class TDM:
    def __init__(self, engine: Engine, model_name="din"):
        self.engine = engine
        self.use_mask = model_name.lower() == "din"   # TDM.apply, TDM.scala:26-29

    def predict(self, sequence, target):
        """TDM.predict(sequence, target): Double (TDM.scala:10-15): the model's probability for ONE (history, target item) pair —
        idToCode over sequence ++ target, one forward, sigmoid in double.  (The reference hands the model a single [1, L + 1] tensor,
        which only its DeepFM graph accepts; for DIN the same row goes through Module.forward(Table(item, sequence, mask)).)"""
        seq = np.asarray(sequence, dtype=np.int32).ravel()
        codes, mask_pos = self.engine.id_to_code(np.concatenate([seq, np.asarray([target], np.int32)]))
        pad = mask_pos[mask_pos < seq.size].astype(np.int32) if self.use_mask else np.zeros(0, np.int32)
        logit = self.engine.din_forward(codes[-1:], codes[None, :-1], pad)
        return float(sigmoid(np.float32(logit[0])))

    @staticmethod
    def load_tree(engine, tree_pb_path):
        """TDM.loadTree(treePbPath) (TDM.scala:50-52 -> TDMOp.initTree): the reference's tree file into the engine."""
        engine.load_tree_file(tree_pb_path)

    def recommend(self, sequence, topk, candidate_num):
        """TDM.recommend(sequence, topk, candidateNum): Array[(Int, Double)] (TDM.scala:17-22)."""
        seq = np.asarray(sequence, dtype=np.int32)
        single = seq.ndim == 1
        ids, sc, cnt = self.engine.tdm_beam_search(seq, candidate_num, topk, use_mask=self.use_mask)
        prob = sigmoid(sc)                                   # one vectorised pass, in double
        out = [list(zip(ids[u, :cnt[u]].tolist(), prob[u, :cnt[u]].tolist())) for u in range(ids.shape[0])]
        return out[0] if single else out

    def save_model(self, model_path):
        """TDM.saveModel (TDM.scala:32-41): one flat checkpoint instead of a Java-serialised module graph."""
        self.engine.save_model(model_path)

    @classmethod
    def load_model(cls, engine, path, model_name="din"):
        """TDM.loadModel (TDM.scala:43-54)."""
        engine.load_model(path)
        return cls(engine, model_name)

    def recommend_items(self, sequence, topk, candidate_num, consumed_items=None):
        """Recommender.recommendItems (Recommender.scala:18-37): ids only, beam widened by consumed count."""
        seq = np.asarray(sequence, dtype=np.int32)
        single = seq.ndim == 1
        consumed = None
        if consumed_items is not None:
            consumed = [consumed_items] if single else consumed_items
        ids, _, cnt = self.engine.tdm_beam_search(seq, candidate_num, topk, use_mask=self.use_mask, consumed=consumed,
                                                  widen_consumed=consumed is not None)
        out = [ids[u, :cnt[u]].copy() for u in range(ids.shape[0])]
        return out[0] if single else out


class OTM:
    def __init__(self, engine: Engine, item_id_mapping, model_name="din"):
        """item_id_mapping: dict item -> leaf node id (OTM.scala:8-12)."""
        self.engine = engine
        self.item_id_mapping = dict(item_id_mapping)
        self.id_item_mapping = {v: k for k, v in self.item_id_mapping.items()}
        n = len(self.item_id_mapping)
        self.leaf_level = int(np.ceil(np.log(n) / np.log(2)))     # upperLog2, otm/package.scala:16
        self.use_mask = model_name.lower() == "din"
        size = (1 << (self.leaf_level + 1)) - 1
        self._node_to_item = np.full(size, -1, np.int32)
        for item, node in self.item_id_mapping.items():
            if 0 <= node < size:
                self._node_to_item[node] = item

    def recommend(self, sequence, topk, beam_size):
        """OTM.recommend(sequence, topk, beamSize): Seq[(Int, Double)] (OTM.scala:14-22)."""
        seq = np.asarray(sequence, dtype=np.int64)
        single = seq.ndim == 1
        if single:
            seq = seq[None, :]
        get = self.item_id_mapping.get
        codes = np.array([[get(i, -1) for i in row] for row in seq.tolist()], dtype=np.int32)
        ids, sc, cnt = self.engine.otm_beam_search(codes, beam_size, self.leaf_level)
        out = []
        n2i = self._node_to_item
        for u in range(ids.shape[0]):
            nodes, scores = ids[u, :cnt[u]], sc[u, :cnt[u]]
            items = np.where((nodes >= 0) & (nodes < n2i.size), n2i[np.clip(nodes, 0, n2i.size - 1)], -1)
            keep = items >= 0                                  # filter(idItemMapping.contains)
            items, scores = items[keep], scores[keep]
            order = np.argsort(-scores, kind="stable")[:topk]  # stable sort descending by score (sortBy(_.score)(Ordering[Double].reverse))
            out.append(list(zip(items[order].tolist(), sigmoid(scores[order]).tolist())))
        return out[0] if single else out


class DeepRetrieval:
    """DeepRetrieval.recommend(sequence, topk, beamSize, mappings): Seq[(Int, Double)]
    (deep-retrieval/src/main/scala/com/mass/dr/model/DeepRetrieval.scala:26-46).

    item_id_mapping: dict item -> internal id (MappingOp.itemIdMapping); the path -> items table is loaded into the
    engine separately (Engine.dr_load_path_items)."""

    def __init__(self, engine: Engine, item_id_mapping):
        self.engine = engine
        self.item_id_mapping = dict(item_id_mapping)
        self.id_item_mapping = {v: k for k, v in self.item_id_mapping.items()}     # MappingOp.scala:16

    def recommend(self, sequence, topk, beam_size):
        seq = np.asarray(sequence, dtype=np.int64)
        single = seq.ndim == 1
        if single:
            seq = seq[None, :]
        get = self.item_id_mapping.get
        ids = np.array([[get(i, -1) for i in row] for row in seq.tolist()], dtype=np.int32)   # :32
        out_ids, sc, cnt = self.engine.dr_recommend(ids, beam_size, topk)
        back = self.id_item_mapping
        prob = sigmoid(sc)
        out = [[(back[i], pr) for i, pr in zip(out_ids[u, :cnt[u]].tolist(), prob[u, :cnt[u]].tolist())] for u in range(ids.shape[0])]
        return out[0] if single else out
