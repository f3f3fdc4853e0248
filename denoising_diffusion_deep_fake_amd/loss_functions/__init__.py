from .structural_similarity_loss import MseStructuralSimilarityLoss  # noqa: F401
