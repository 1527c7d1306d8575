"""Placeholder import target; the nn.Module mirror of the reference's mfm_model.py lands here."""
