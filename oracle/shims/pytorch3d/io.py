"""Import shim (build container only): mesh IO is never reached on the hot path."""
