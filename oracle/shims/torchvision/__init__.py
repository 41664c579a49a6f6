"""Import shim (build container only)."""
