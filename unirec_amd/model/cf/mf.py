"""MF -- mirror of unirec/model/cf/mf.py:7-11: user_emb = user_embedding(user_id) (recommender.py:42-44)."""
from ..base.recommender import BaseRecommender


class MF(BaseRecommender):
    def add_annotation(self):
        super().add_annotation()
        self.annotations.append("MF")

    def _define_model_layers(self):
        if not self.config["has_user_emb"]:
            raise ValueError("MF needs has_user_emb=True (unirec/config/model/MF.yaml:3)")
        self._alloc_dense(4)  # MF has no dense parameters; keep a tiny buffer so the optimizer path is uniform
