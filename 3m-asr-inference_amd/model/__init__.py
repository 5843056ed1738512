"""Model descriptions (L1 of the reference: trainer_3m_fix/model).  A model's ``encoder(network_helper, feat,
feat_len)`` emits the encoder through the graph-builder API exactly as the reference's forward() bodies do."""
