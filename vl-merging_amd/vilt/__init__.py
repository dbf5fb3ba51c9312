"""Host-side mirror of the reference's `vilt` package for the hot path (same module / parameter names)."""
