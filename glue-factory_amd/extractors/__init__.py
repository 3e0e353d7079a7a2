"""Feature extractors on the measured pipeline path (stock PyTorch-ROCm ops by design)."""
