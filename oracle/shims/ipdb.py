"""Import shim (build container only): the reference imports ipdb everywhere; a live set_trace() must fail loudly."""
def set_trace(*a, **k):
    raise RuntimeError("ipdb.set_trace() reached inside the reference (shim)")
